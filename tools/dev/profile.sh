#!/bin/bash
# Runs on the GPU box (via gpurun): for each workload named on the command line (default: all six of bench.py)
#   <w>/stats  rocprofv3 --kernel-trace --stats of `bench.py --workload <w>` (kernel statistics + the bench line of the traced run)
#   <w>/fetch, <w>/write  HBM byte counters, separate passes (FETCH_SIZE and WRITE_SIZE do not fit one)
# plus, once: a calibration of the byte counters on kernels with known traffic per access pattern (zg_k_calib_*), and for the
# first workload the SQ counters of the two LZ77 kernels (raw CSVs, one row per launch).
# Writes under gpurun_out/prof; tools/dev/pmc_summary.py turns that into profiles/r06/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
WL=${*:-"enwik9like realtext1g blocks blocks4b iso silesia12 realtext"}
B="--no-cpu --no-e2e --no-other --min-seconds 0"
first=1
for w in $WL; do
  mkdir -p $OUT/$w
  timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/$w/stats -o trace -- python $ROOT/bench.py --workload $w --steps 10 --warmup 2 $B > $OUT/$w/stats.log 2>&1
  timeout 600 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/$w/fetch -o pmc -- python $ROOT/bench.py --workload $w --steps 2 --warmup 1 $B > $OUT/$w/fetch.log 2>&1
  timeout 600 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/$w/write -o pmc -- python $ROOT/bench.py --workload $w --steps 2 --warmup 1 $B > $OUT/$w/write.log 2>&1
  if [ "${SQ_ALL:-1}" = 1 ] || [ $first = 1 ]; then   # SQ counters (instruction mix, busy / wait cycles, LDS bank conflicts): every workload unless SQ_ALL=0
    first=0
    k=1
    for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"; do
      timeout 600 rocprofv3 --output-format csv --pmc $set -d $OUT/$w/sq$k -o pmc -- python $ROOT/bench.py --workload $w --steps 1 --warmup 1 $B > $OUT/$w/sq$k.log 2>&1
      k=$((k+1))
    done
  fi
done
cat > /tmp/calib.py <<PY
import sys
sys.path.insert(0, "$ROOT/zstd-rs_amd")
import zgpu
c = zgpu.Context(0)
assert c.L.zgpu_debug_calibrate(c.h, 1 << 30) == 0
PY
timeout 300 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/calib_fetch -o pmc -- python /tmp/calib.py > $OUT/calib_fetch.log 2>&1
timeout 300 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/calib_write -o pmc -- python /tmp/calib.py > $OUT/calib_write.log 2>&1
# keep what the summary needs, drop the bulky traces
python $ROOT/tools/dev/pmc_summary.py --reduce $WL
find $OUT -name "*.csv" -size +3M -delete
for w in $WL; do tail -1 $OUT/$w/stats.log | cut -c1-300; done

#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace statistics and HBM byte counters of bench.py (separate passes: FETCH_SIZE and
# WRITE_SIZE do not fit one), plus a calibration of the byte counters on kernels with known traffic per access pattern.
# Writes under gpurun_out/prof (tools/dev/pmc_summary.py turns that into profiles/r02/).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS=${*:-"--steps 10 --warmup 2 --no-cpu --no-e2e --no-other"}
timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -o trace -- python $ROOT/bench.py $ARGS > $OUT/stats.log 2>&1
timeout 600 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/fetch -o pmc -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-e2e --no-other > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/write -o pmc -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-e2e --no-other > $OUT/write.log 2>&1
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
python $ROOT/tools/dev/pmc_summary.py --reduce
find $OUT -name "*.csv" -size +2M -delete
tail -2 $OUT/stats.log | cut -c1-400

#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace statistics and HBM byte counters of bench.py, plus a calibration of
# the byte counters on a copy kernel with known traffic. Writes under gpurun_out/ (copied into profiles/ afterwards).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SIZE=${1:-1000000000}
timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -o trace -- python $ROOT/bench.py --size $SIZE --steps 3 --warmup 1 --no-cpu > $OUT/stats.log 2>&1
timeout 600 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/fetch -o pmc -- python $ROOT/bench.py --size $SIZE --steps 1 --warmup 1 --no-cpu > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/write -o pmc -- python $ROOT/bench.py --size $SIZE --steps 1 --warmup 1 --no-cpu > $OUT/write.log 2>&1
cat > /tmp/calib.py <<PY
import sys
sys.path.insert(0, "$ROOT/zstd-rs_amd")
import zgpu
c = zgpu.Context(0)
assert c.L.zgpu_debug_calibrate(c.h, 1 << 30) == 0
PY
timeout 300 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/calib_fetch -o pmc -- python /tmp/calib.py > $OUT/calib_fetch.log 2>&1
timeout 300 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/calib_write -o pmc -- python /tmp/calib.py > $OUT/calib_write.log 2>&1
find $OUT -name "*.csv" | head -30
tail -2 $OUT/stats.log

#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c26
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
for j in 1 2 4 8; do
  ZGPU_POOL_JOBS=$j timeout 600 python bench.py --workload blocks4b --no-cpu --no-e2e --no-other --steps 3 --warmup 1 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('jobs $j blocks4b', d['value'], d['config']['ms_per_pass'], d['per_gpu_busy_ms'], {k:round(v,2) for k,v in d['kernel_ms'].items()})" >> $OUT/jobs.log 2>&1
done
for j in 2 4; do
  ZGPU_POOL_JOBS=$j timeout 600 python bench.py --workload blocks --no-cpu --no-e2e --no-other --steps 3 --warmup 1 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('jobs $j blocks', d['value'], d['config']['ms_per_pass'])" >> $OUT/jobs.log 2>&1
done
cat $OUT/jobs.log | cut -c1-300

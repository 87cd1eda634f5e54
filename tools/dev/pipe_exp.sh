#!/bin/bash
# the jobs of a GPU as a pipeline (ZGPU_POOL_PIPE): bench workloads with 1 / 2 / 4 / 8 jobs, pipelined or side by side
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/pipe_exp.log; : > $O
B="--no-cpu --no-e2e --no-other"
for w in ${*:-blocks}; do
for v in "ZGPU_POOL_JOBS=1" "ZGPU_POOL_JOBS=2 ZGPU_POOL_PIPE=1" "ZGPU_POOL_JOBS=4 ZGPU_POOL_PIPE=1" "ZGPU_POOL_JOBS=8 ZGPU_POOL_PIPE=1" "ZGPU_POOL_JOBS=2"; do
  echo "== $w $v" >> $O
  env $v timeout 600 python bench.py --workload $w $B 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d.get('per_gpu_busy_ms'), d.get('kernel_ms'))" >> $O 2>&1
done
done
cat $O

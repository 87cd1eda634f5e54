#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/c8; mkdir -p $O
cd $ROOT
( time timeout 900 python -m pytest tests/test_gpu_scale.py -x -q -m gpu -k "pool" ) > $O/tests.log 2>&1
tail -5 $O/tests.log
for j in 1 2 4 8; do
  for w in blocks blocks4b iso; do
    ZGPU_POOL_JOBS=$j python bench.py --workload $w --steps 3 --min-seconds 1.0 --no-cpu --no-e2e --no-other 2>$O/b_${w}_$j.err | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$w jobs $j', d['value'], 'GB/s', d['config']['ms_per_pass'], 'ms', {k: round(v, 1) for k, v in d['kernel_ms'].items()})
"
  done
done

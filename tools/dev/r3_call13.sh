#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/c13; mkdir -p $O
cd $ROOT
timeout 300 python tools/dev/e2e.py -- "" ZGPU_DA_SPLIT=16 "ZGPU_DA_SPLIT=16 ZGPU_DA_FLOOR_MB=16" "ZGPU_DA_SPLIT=4" > $O/e2e.log 2>&1; cat $O/e2e.log
timeout 600 python -m pytest tests/test_gpu_scale.py tests/test_gpu_exact.py -x -q -k "pool or exact or streamed or dictionary or decide" > $O/t.log 2>&1; tail -3 $O/t.log

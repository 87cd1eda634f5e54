#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/c11; mkdir -p $O
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_exact.py tests/test_gpu_capi_client.py -x -q > $O/exact.log 2>&1; tail -15 $O/exact.log
timeout 900 python -m pytest tests -x -q -m gpu > $O/all.log 2>&1; tail -5 $O/all.log

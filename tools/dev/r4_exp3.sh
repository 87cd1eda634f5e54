#!/bin/bash
# round 4, GPU call 3: flat1 micro-optimizations (resource-bounded predication), all GPU tests
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > gpurun_out/exp3_tests.log 2>&1
( timeout 600 python tools/dev/variants.py 1000000000 text -- "" ZGPU_FLAT_MODE=3 ) > gpurun_out/exp3_text.log 2>&1
( timeout 900 python tools/dev/variants.py 8589934592 many -- "" ) > gpurun_out/exp3_many.log 2>&1
cat gpurun_out/exp3_tests.log gpurun_out/exp3_text.log gpurun_out/exp3_many.log

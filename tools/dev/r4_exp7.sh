#!/bin/bash
# round 4, GPU call 7: flat4 with the byte-strided S2
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/exp7_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/exp7_tests.log
grep -v "^  File" gpurun_out/exp7_tests.log | tail -8
( timeout 900 python tools/dev/variants.py 2147483648 blocks -- "" ) > gpurun_out/exp7_blocks.log 2>&1
( timeout 900 python tools/dev/variants.py 8589934592 many -- "" ) > gpurun_out/exp7_many.log 2>&1
cat gpurun_out/exp7_blocks.log gpurun_out/exp7_many.log

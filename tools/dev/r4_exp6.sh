#!/bin/bash
# round 4, GPU call 6: all GPU tests with the full log (a run aborted at session end), the bench line
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/exp6_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/exp6_tests.log
grep -v "^  File" gpurun_out/exp6_tests.log | tail -30
( timeout 900 python tools/dev/variants.py 2147483648 blocks -- "" ) > gpurun_out/exp6_blocks.log 2>&1
cat gpurun_out/exp6_blocks.log

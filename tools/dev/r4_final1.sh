#!/bin/bash
# round 4: the default bench line + the GPU suite + smoke, as the driver runs them
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
( time timeout 1500 python bench.py ) > gpurun_out/final_bench.log 2>&1
tail -c 600 gpurun_out/final_bench.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final_smoke.log 2>&1; tail -2 gpurun_out/final_smoke.log
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/final_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/final_tests.log
grep -v "^  File" gpurun_out/final_tests.log | tail -4

#!/bin/bash
# round 4, GPU call 10: zg_k_huf with the trimmed step and a warm-up loop of its own
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/exp10_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/exp10_tests.log
grep -v "^  File" gpurun_out/exp10_tests.log | tail -6
( timeout 900 python tools/dev/variants.py 8589934592 isomany -- "" ) > gpurun_out/exp10_iso.log 2>&1
( timeout 600 python tools/dev/variants.py 1000000000 text -- "" ) > gpurun_out/exp10_text.log 2>&1
cat gpurun_out/exp10_iso.log gpurun_out/exp10_text.log

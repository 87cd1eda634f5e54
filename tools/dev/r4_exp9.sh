#!/bin/bash
# round 4, GPU call 9: literal runs placed by zg_k_litrun beside the flatten
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/exp9_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/exp9_tests.log
grep -v "^  File" gpurun_out/exp9_tests.log | tail -8
( timeout 600 python tools/dev/variants.py 1000000000 text -- "" ) > gpurun_out/exp9_text.log 2>&1
( timeout 900 python tools/dev/variants.py 8589934592 many -- "" ) > gpurun_out/exp9_many.log 2>&1
( timeout 900 python tools/dev/variants.py 8589934592 isomany -- "" ) > gpurun_out/exp9_iso.log 2>&1
( timeout 900 python tools/dev/configs.py silesia ) > gpurun_out/exp9_silesia.log 2>&1
cat gpurun_out/exp9_text.log gpurun_out/exp9_many.log gpurun_out/exp9_iso.log gpurun_out/exp9_silesia.log

#!/bin/bash
# round 4, GPU call 8: direct first units with a 1.3x share
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/exp8_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/exp8_tests.log
grep -v "^  File" gpurun_out/exp8_tests.log | tail -8
( timeout 900 python tools/dev/variants.py 8589934592 many -- "" ) > gpurun_out/exp8_many.log 2>&1
( timeout 900 python tools/dev/configs.py silesia ) > gpurun_out/exp8_silesia.log 2>&1
cat gpurun_out/exp8_many.log gpurun_out/exp8_silesia.log

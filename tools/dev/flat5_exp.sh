#!/bin/bash
# zg_flat5.h (pointer-mode flatten at dword granularity, byte-strided jumping) against zg_flat1.h
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
( timeout 600 python tools/dev/variants.py 1000000000 text -- "" ZGPU_FLAT5=1 "" ZGPU_FLAT5=1 ) > gpurun_out/flat5_text.log 2>&1
( timeout 900 python tools/dev/variants.py 4294967296 many -- "" ZGPU_FLAT5=1 ) > gpurun_out/flat5_many.log 2>&1
ZGPU_FLAT5=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/flat5_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/flat5_tests.log
cat gpurun_out/flat5_text.log gpurun_out/flat5_many.log; grep -v "^  File" gpurun_out/flat5_tests.log | tail -5

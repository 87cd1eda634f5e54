#!/bin/bash
# round 4, GPU call 11: zg_k_huf with packed symbol stores; the direct unit's share
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/exp11_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/exp11_tests.log
grep -v "^  File" gpurun_out/exp11_tests.log | tail -6
( timeout 900 python tools/dev/variants.py 8589934592 isomany -- "" ) > gpurun_out/exp11_iso.log 2>&1
( timeout 900 python tools/dev/variants.py 8589934592 many -- "" ZGPU_DIRECT_SHARE=15 ZGPU_DIRECT_SHARE=17 ) > gpurun_out/exp11_many.log 2>&1
cat gpurun_out/exp11_iso.log gpurun_out/exp11_many.log

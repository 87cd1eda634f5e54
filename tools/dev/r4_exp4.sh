#!/bin/bash
# round 4, GPU call 4: zg_k_huf with LDS-staged 16-byte output stores
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > gpurun_out/exp4_tests.log 2>&1
( timeout 900 python tools/dev/variants.py 8589934592 isomany -- "" ) > gpurun_out/exp4_iso.log 2>&1
( timeout 600 python tools/dev/variants.py 1000000000 text -- "" ) > gpurun_out/exp4_text.log 2>&1
cat gpurun_out/exp4_tests.log gpurun_out/exp4_iso.log gpurun_out/exp4_text.log

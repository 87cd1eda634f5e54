#!/bin/bash
# round 4, GPU call 2: the new zg_k_huf (register window) and the literals-after-the-scan order
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/exp2_tests.log 2>&1
( timeout 900 python tools/dev/variants.py 8589934592 isomany -- "" ZGPU_LIT_DIRECT=0 ) > gpurun_out/exp2_iso.log 2>&1
( timeout 600 python tools/dev/variants.py 1000000000 text -- "" ) > gpurun_out/exp2_text.log 2>&1
( timeout 600 python tools/dev/variants.py 268435456 blocks -- "" ) > gpurun_out/exp2_blocks.log 2>&1
cat gpurun_out/exp2_tests.log gpurun_out/exp2_iso.log gpurun_out/exp2_text.log gpurun_out/exp2_blocks.log

#!/bin/bash
# round 4, GPU call 5: 24-bit scratch words (ZG_FLAG_OG24), group-wise S3b, pre-sized output
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > gpurun_out/exp5_tests.log 2>&1
( timeout 600 python tools/dev/variants.py 1000000000 text -- "" ZGPU_OG24=0 ZGPU_PRESIZE=0 "" ) > gpurun_out/exp5_text.log 2>&1
( timeout 900 python tools/dev/variants.py 8589934592 many -- "" ZGPU_OG24=0 ) > gpurun_out/exp5_many.log 2>&1
cat gpurun_out/exp5_tests.log gpurun_out/exp5_text.log gpurun_out/exp5_many.log

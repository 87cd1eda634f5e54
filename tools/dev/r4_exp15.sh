#!/bin/bash
# round 4, GPU call 15: what the literal bytes' loads and stores cost in S3b of zg_flat1_unit (timing mode 4: without them, wrong bytes)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
( timeout 600 python tools/dev/variants.py 1000000000 text -- "" ZGPU_FLAT_MODE=4 ZGPU_FLAT_MODE=7 "" ) > gpurun_out/exp15_text.log 2>&1
( timeout 900 python tools/dev/variants.py 4294967296 many -- "" ZGPU_FLAT_MODE=4 ) > gpurun_out/exp15_many.log 2>&1
cat gpurun_out/exp15_text.log gpurun_out/exp15_many.log

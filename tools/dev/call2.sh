#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c25
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 600 python $ROOT/tools/dev/variants.py 1000000000 text -- "" ZGPU_SWEEP_HEAD_LDS=81920 ZGPU_SWEEP_HEAD_LDS=110000 ZGPU_SWEEP_HEAD_LDS=30000 ZGPU_SWEEP_HEAD_LDS=81920,ZGPU_SWEEP_GROUP=32 ZGPU_SWEEP_HEAD_LDS=81920,ZGPU_SWEEP_GROUP=8 > $OUT/text.log 2>&1
cat $OUT/text.log | cut -c1-330

#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c8
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 600 python $ROOT/tools/dev/variants.py 2147483648 blocks -- ZGPU_FLAT4=0 ZGPU_FLAT4=4 ZGPU_FLAT4=5 ZGPU_FLAT4=6 ZGPU_FLAT4=7 ZGPU_FLAT4=1 > $OUT/blocks.log 2>&1
cat $OUT/blocks.log | cut -c1-330

#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c10
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
( ZGPU_SEQ_PACKED=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu ) > $OUT/tests_packed.log 2>&1
tail -3 $OUT/tests_packed.log
cd /tmp
timeout 600 python $ROOT/tools/dev/variants.py 2147483648 blocks -- ZGPU_SEQ_PACKED=0 ZGPU_SEQ_PACKED=1 > $OUT/blocks.log 2>&1
timeout 600 python $ROOT/tools/dev/variants.py 4294967296 many -- ZGPU_SEQ_PACKED=0 ZGPU_SEQ_PACKED=1 > $OUT/many.log 2>&1
timeout 600 python $ROOT/tools/dev/variants.py 1000000000 text -- ZGPU_SEQ_PACKED=0 ZGPU_SEQ_PACKED=1 > $OUT/text.log 2>&1
cat $OUT/blocks.log $OUT/many.log $OUT/text.log | cut -c1-330

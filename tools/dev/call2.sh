#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c23
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 300 python $ROOT/tools/dev/configs.py silesia many > $OUT/silesia.log 2>&1
timeout 600 python $ROOT/tools/dev/variants.py 4294967296 many >> $OUT/silesia.log 2>&1
cat $OUT/silesia.log | cut -c1-330
cd $ROOT; ( timeout 900 python -m pytest tests -x -q -m gpu ) > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log

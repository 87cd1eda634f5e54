#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c16
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -x -q -m gpu ) > $OUT/tests.log 2>&1
tail -15 $OUT/tests.log

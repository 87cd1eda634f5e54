#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c17
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -x -q -m gpu -k "packed_sequence or 640_multi" ) > $OUT/tests.log 2>&1
tail -15 $OUT/tests.log

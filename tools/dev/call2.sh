#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c19
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_thin_boundary.py -x -q -m gpu ) > $OUT/tests.log 2>&1
tail -12 $OUT/tests.log

#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c5
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
( time timeout 500 $ROOT/tools/dev/chain_fill_bench 255 ) > $OUT/chain_fill_bench.log 2>&1
cat $OUT/chain_fill_bench.log

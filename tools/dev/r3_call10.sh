#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/c10; mkdir -p $O
cd $ROOT
timeout 300 python tools/dev/variants.py 1000000000 text -- ZGPU_RAMP=0 "" ZGPU_RAMP=30 ZGPU_RAMP=70 > $O/var_text.log 2>&1; cat $O/var_text.log

#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/c9; mkdir -p $O
cd $ROOT
timeout 300 python tools/dev/variants.py 1000000000 text -- "" ZGPU_OVERLAP=0 ZGPU_RAMP=0 > $O/var_text.log 2>&1; cat $O/var_text.log
timeout 300 python tools/dev/variants.py 1000000000 text -- ZGPU_RAMP=30 ZGPU_RAMP=70 ZGPU_RAMP=90 > $O/var_text1.log 2>&1; cat $O/var_text1.log
( time timeout 1700 python -m pytest tests -x -q -m gpu ) > $O/gputests.log 2>&1
tail -5 $O/gputests.log

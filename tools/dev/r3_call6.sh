#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/c6; mkdir -p $O
cd $ROOT
( time timeout 1700 python -m pytest tests -x -q -m gpu --durations=15 ) > $O/gputests.log 2>&1
tail -30 $O/gputests.log
python tools/dev/variants.py 1000000000 text -- "" ZGPU_DIRECT=0 > $O/var_text.log 2>&1; cat $O/var_text.log
python tools/dev/variants.py 268435456 blocks -- "" ZGPU_DIRECT=0 > $O/var_blocks.log 2>&1; cat $O/var_blocks.log
python tools/dev/configs.py many silesia iso > $O/cfg_new.log 2>&1
ZGPU_DIRECT=0 python tools/dev/configs.py many silesia iso > $O/cfg_old.log 2>&1
cat $O/cfg_new.log $O/cfg_old.log

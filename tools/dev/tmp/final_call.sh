#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/c14; mkdir -p $O
cd $ROOT
timeout 900 python -m pytest tests -x -q -m gpu > $O/all.log 2>&1; tail -3 $O/all.log
timeout 900 python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-600
bash tools/dev/profile.sh > $O/profile.log 2>&1; tail -6 $O/profile.log | cut -c1-200

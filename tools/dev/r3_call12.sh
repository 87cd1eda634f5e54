#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/c12; mkdir -p $O
cd $ROOT
timeout 900 python -m pytest tests -x -q -m gpu --durations=8 > $O/all.log 2>&1; tail -16 $O/all.log
timeout 600 python bench.py --no-other > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-1500

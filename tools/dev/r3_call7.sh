#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/c7; mkdir -p $O
cd $ROOT
( time timeout 900 python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py -x -q -m gpu -k "config4 or pool_over or streaming_window or pool_queue" ) > $O/tests.log 2>&1
tail -15 $O/tests.log
( time python bench.py --steps 5 ) > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err; cat $O/bench.json | cut -c1-6000
( time python bench.py --gpus 2 --steps 3 --no-other --no-cpu --no-e2e ) > $O/bench_g2.json 2> $O/bench_g2.err
tail -3 $O/bench_g2.err; cat $O/bench_g2.json | cut -c1-1500

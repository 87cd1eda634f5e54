#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT; mkdir -p gpurun_out/c24
( time timeout 1500 python bench.py ) > gpurun_out/c24/bench.log 2> gpurun_out/c24/bench.err
tail -3 gpurun_out/c24/bench.err

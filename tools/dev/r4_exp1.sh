#!/bin/bash
# round 4, GPU call 1: parity of the changed kernels, then what the LZ77 scratch traffic is worth (timing modes: wrong results on purpose)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
export ZGPU_SWEEP_MODE_OK=1
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/exp1_tests.log 2>&1
( timeout 600 python tools/dev/variants.py 1000000000 text -- "" ZGPU_FLAT_MODE=1 ZGPU_FLAT_MODE=2 ZGPU_FLAT_MODE=3 ZGPU_SWEEP_MODE=5 ZGPU_SWEEP_MODE=6 ZGPU_SWEEP_MODE=7 ZGPU_SWEEP_MODE=8 ZGPU_SWEEP_MODE=2 ZGPU_FLAT_T=512 ZGPU_FLAT_T=512,ZGPU_UNIT_BLOCKS=15 "" ) > gpurun_out/exp1_text.log 2>&1
( timeout 900 python tools/dev/variants.py 8589934592 many -- "" ZGPU_FLAT_T=512 ZGPU_FLAT_T=512,ZGPU_UNIT_BLOCKS=128 ZGPU_UNIT_BLOCKS=128 ZGPU_UNIT_BLOCKS=64 ZGPU_FLAT_T=512,ZGPU_UNIT_BLOCKS=64 ) > gpurun_out/exp1_many.log 2>&1
cat gpurun_out/exp1_tests.log gpurun_out/exp1_text.log gpurun_out/exp1_many.log

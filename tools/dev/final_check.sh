#!/bin/bash
# after a late kernel change: differential soak, the GPU suite, the headline workload's profile, the bench line
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
SOAK_VARIANTS=";" SOAK_SEED=7 SOAK_PER=30 bash tools/dev/soak.sh 2>&1 | head -30
python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^  File" | tail -4
bash tools/dev/profile.sh enwik9like > gpurun_out/profile_sh.log 2>&1; tail -2 gpurun_out/profile_sh.log | cut -c1-200
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_final.log 2>&1; tail -c 300 gpurun_out/bench_final.log

#!/bin/bash
# after a late change: every differential soak (fixed seeds, a few minutes together), the GPU suite, and — when a kernel source changed
# (bench.py: kernels_sha256) — the profiles of all workloads + the bench line.   usage (on the GPU box): bash tools/dev/final_check.sh [profile]
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
S=${SOAK_SEED:-7}
(
timeout 300 python tools/dev/soak.py 25 $S | tail -2
timeout 400 python tools/dev/soak_stream.py 3000 $S | tail -1
timeout 300 python tools/dev/soak_api.py 6000 $S | tail -1
timeout 300 python tools/dev/soak_concat.py 2000 $S | tail -1
timeout 300 python tools/dev/soak_thin.py 3000 $S | tail -1
timeout 300 python tools/dev/soak_batch.py 2000 $S | tail -1
timeout 400 python tools/dev/soak_seqbits.py 2000 $S | tail -1
timeout 300 python tools/dev/soak_headers.py 1500 $S | tail -1
timeout 400 python tools/dev/soak_pool.py 100 $S | tail -1
timeout 600 python tools/dev/soak_big.py 6 $S 200,640 | tail -1
) 2>&1 | grep -v amdgpu.ids | cut -c1-240
python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^  File" | tail -3
if [ "${1:-}" = profile ]; then
  SQ_ALL=0 bash tools/dev/profile.sh > gpurun_out/profile_sh.log 2>&1; tail -1 gpurun_out/profile_sh.log | cut -c1-200
  python3 bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -c 300 gpurun_out/bench_final.json
fi

#!/bin/bash
# end of a round: profiles of every workload (tools/dev/profile.sh), then the default bench line
cd ${GRAFT_REPO_ROOT:-.}
bash tools/dev/profile.sh > gpurun_out/profile_sh.log 2>&1
tail -8 gpurun_out/profile_sh.log | cut -c1-200
( time timeout 1500 python bench.py ) > gpurun_out/final_bench.log 2>&1
tail -c 300 gpurun_out/final_bench.log

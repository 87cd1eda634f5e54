#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash tools/dev/profile.sh silesia12 > gpurun_out/profile_silesia.log 2>&1
tail -3 gpurun_out/profile_silesia.log | cut -c1-300

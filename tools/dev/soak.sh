#!/bin/bash
# differential soak (tools/dev/soak.py) under the engine's path switches; SOAK_PER mutations per frame, SOAK_SEED, SOAK_VARIANTS
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
: > gpurun_out/soak.log
IFS=';' read -ra VARS <<< "${SOAK_VARIANTS:-;ZGPU_LIT_DIRECT=1;ZGPU_SPARSE_MAX=100000000;ZGPU_PRESIZE=0,ZGPU_LIT_DIRECT=0;ZGPU_DIRECT=0;ZGPU_UNIT_BLOCKS=1}"
[ ${#VARS[@]} -eq 0 ] && VARS=("")
for v in "${VARS[@]}"; do
  timeout 900 python tools/dev/soak.py ${SOAK_PER:-30} ${SOAK_SEED:-7} "$v" >> gpurun_out/soak.log 2>&1
done
cut -c1-300 gpurun_out/soak.log | head -80

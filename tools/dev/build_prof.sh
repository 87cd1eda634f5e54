#!/bin/bash
# profiling build of the library: libzgpu_prof.so with the phase counters of the given kernels compiled in
# usage: tools/dev/build_prof.sh -DZG_PROFILE_FLAT [-DZG_PROFILE_SEQ ...]
set -e
cd "$(dirname "$0")/../../zstd-rs_amd/csrc"
make -s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -x hip zg_kernels.hip -c -o /tmp/zg_kernels_prof.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libzgpu_prof.so /tmp/zg_kernels_prof.o zg_engine.o zg_capi.o zg_pool.o zg_host_parse.o -lpthread
echo built ../libzgpu_prof.so

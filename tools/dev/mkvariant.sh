#!/bin/bash
# a variant build of libzgpu.so for measurements: mkvariant.sh NAME "-DZG_X=.." -> zstd-rs_amd/libzgpu_NAME.so (use with ZGPU_LIB=...)
set -e
cd "$(dirname "$0")/../../zstd-rs_amd/csrc"
make -s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $2 -x hip zg_kernels.hip -c -o /tmp/zg_kernels_$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libzgpu_$1.so /tmp/zg_kernels_$1.o zg_engine.o zg_capi.o zg_pool.o zg_stream.o zg_host_parse.o -lpthread
echo built ../libzgpu_$1.so

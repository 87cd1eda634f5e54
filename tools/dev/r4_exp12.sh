#!/bin/bash
# round 4, GPU call 12: literal runs of sparse frames without tiles; zg_k_huf warm-up 32 / 64 / 96 bits
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/exp12_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/exp12_tests.log
grep -v "^  File" gpurun_out/exp12_tests.log | tail -6
( timeout 600 python tools/dev/variants.py 4294967296 isomany -- "" ZGPU_LIT_DIRECT=0 ) > gpurun_out/exp12_iso.log 2>&1
( ZGPU_LIB=$PWD/zstd-rs_amd/libzgpu_warm64.so timeout 600 python tools/dev/variants.py 4294967296 isomany -- "" | sed 's/^default/warm64 /' ) >> gpurun_out/exp12_iso.log 2>&1
( ZGPU_LIB=$PWD/zstd-rs_amd/libzgpu_warm96.so timeout 600 python tools/dev/variants.py 4294967296 isomany -- "" | sed 's/^default/warm96 /' ) >> gpurun_out/exp12_iso.log 2>&1
( timeout 600 python tools/dev/variants.py 1000000000 text -- "" ) > gpurun_out/exp12_text.log 2>&1
cat gpurun_out/exp12_iso.log gpurun_out/exp12_text.log

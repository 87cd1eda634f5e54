#!/bin/bash
# round 4, GPU call 14: zg_k_huf with four symbols per dword in LDS and a dword per iteration of the output loop, against the byte form
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/exp14_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/exp14_tests.log
grep -v "^  File" gpurun_out/exp14_tests.log | tail -6
( timeout 600 python tools/dev/variants.py 4294967296 isomany -- "" ) > gpurun_out/exp14_iso.log 2>&1
( ZGPU_LIB=$PWD/zstd-rs_amd/libzgpu_bytesym.so timeout 600 python tools/dev/variants.py 4294967296 isomany -- "" | sed 's/^default/bytesym/' ) >> gpurun_out/exp14_iso.log 2>&1
( timeout 600 python tools/dev/variants.py 1000000000 text -- "" ) > gpurun_out/exp14_text.log 2>&1
( ZGPU_LIB=$PWD/zstd-rs_amd/libzgpu_bytesym.so timeout 600 python tools/dev/variants.py 1000000000 text -- "" | sed 's/^default/bytesym/' ) >> gpurun_out/exp14_text.log 2>&1
cat gpurun_out/exp14_iso.log gpurun_out/exp14_text.log

#!/bin/bash
# Library with the four-quadrant GEMM kernel (tools/experiments/gemm_q4.hip) as variant 11: afft_amd/lib/libafft_hip_q4.so
#   bash tools/experiments/build_q4.sh [-DAFFT_Q4_DIAG=n] ; AFFT_LIB=$PWD/afft_amd/lib/libafft_hip_q4.so python tools/q4_check.py
set -e
cd "$(dirname "$0")/../../afft_amd/csrc"
make -j8 > /dev/null
mkdir -p build_var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 "$@" -c ../../tools/experiments/gemm_q4.hip -o build_var/gemm_q4.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DAFFT_EXPERIMENT_Q4 -c gemm.hip -o build_var/gemm_with_q4.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libafft_hip_q4.so build_var/gemm_q4.o build_var/gemm_with_q4.o build/gemm_pp.o build/gemm_bd.o build/norm.o build/attention.o build/attention_mfma.o build/loss.o build/elementwise.o build/sublayer.o
ls -la ../lib/libafft_hip_q4.so

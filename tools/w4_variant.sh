#!/bin/bash
# A variant of the library with the four-wave kernels (gemm_w4.hip built with extra -D flags) dispatchable (AFFT_GEMM_VARIANT=5/6):
# tools/w4_variant.sh <name> [-D...] -> afft_amd/lib/libafft_hip_<name>.so ; the other objects from the regular build
set -e
cd "$(dirname "$0")/../afft_amd/csrc"
name=$1; shift
mkdir -p build_var
[ build_var/gemm_exp.o -nt gemm.hip ] && [ build_var/gemm_exp.o -nt gemm_tiles.h ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DAFFT_BUILD_EXPERIMENTAL -c gemm.hip -o build_var/gemm_exp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DAFFT_BUILD_EXPERIMENTAL "$@" -c gemm_w4.hip -o build_var/gemm_w4_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libafft_hip_$name.so build_var/gemm_w4_$name.o build_var/gemm_exp.o build/gemm_pp.o build/norm.o build/attention.o build/attention_mfma.o build/loss.o build/elementwise.o build/sublayer.o

#!/bin/bash
# Builds an experimental variant of the library: tools/pp_build.sh <name> [-DAFFT_PP_...=..] -> afft_amd/lib/libafft_hip_<name>.so
set -e
cd "$(dirname "$0")/../afft_amd/csrc"
name=$1; shift
mkdir -p build_stamp
for f in gemm gemm_w4 norm attention attention_mfma loss elementwise; do
  [ build_stamp/$f.o -nt $f.hip ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -c $f.hip -o build_stamp/$f.o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 "$@" -c gemm_pp.hip -o build_stamp/gemm_pp_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libafft_hip_$name.so build_stamp/gemm_pp_$name.o build_stamp/gemm.o build_stamp/gemm_w4.o build_stamp/norm.o build_stamp/attention.o build_stamp/attention_mfma.o build_stamp/loss.o build_stamp/elementwise.o

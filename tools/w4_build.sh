#!/bin/bash
# Builds an experimental variant of the four-wave GEMM: tools/w4_build.sh <name> [-DAFFT_W4_...=..] -> afft_amd/lib/libafft_hip_<name>.so
set -e
cd "$(dirname "$0")/../afft_amd/csrc"
name=$1; shift
mkdir -p build_stamp
for f in gemm gemm_pp norm attention attention_mfma loss elementwise; do
  [ build_stamp/$f.o -nt $f.hip ] && [ build_stamp/$f.o -nt gemm_tiles.h ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -c $f.hip -o build_stamp/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 "$@" -c gemm_w4.hip -o build_stamp/gemm_w4_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libafft_hip_$name.so build_stamp/gemm_w4_$name.o build_stamp/gemm.o build_stamp/gemm_pp.o build_stamp/norm.o build_stamp/attention.o build_stamp/attention_mfma.o build_stamp/loss.o build_stamp/elementwise.o

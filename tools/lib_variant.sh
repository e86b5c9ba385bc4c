#!/bin/bash
# Builds a variant of the whole library with extra -D flags: tools/lib_variant.sh <name> [-D...] -> afft_amd/lib/libafft_hip_<name>.so
set -e
cd "$(dirname "$0")/../afft_amd/csrc"
name=$1; shift
mkdir -p build_var/$name
for f in gemm gemm_pp gemm_bd norm attention attention_mfma loss elementwise sublayer; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 "$@" -c $f.hip -o build_var/$name/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libafft_hip_$name.so build_var/$name/*.o

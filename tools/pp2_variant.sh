#!/bin/bash
# A variant of the library whose gemm_pp.hip (all instantiations) is built with extra -D flags, the other objects taken from the
# regular build: tools/pp2_variant.sh <name> [-D...] -> afft_amd/lib/libafft_hip_<name>.so   (compare with tools/pp_ab.py)
set -e
cd "$(dirname "$0")/../afft_amd/csrc"
name=$1; shift
mkdir -p build_var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 "$@" -c gemm_pp.hip -o build_var/gemm_pp_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libafft_hip_$name.so build_var/gemm_pp_$name.o build/gemm.o build/gemm_bd.o build/norm.o build/attention.o build/attention_mfma.o build/loss.o build/elementwise.o build/sublayer.o

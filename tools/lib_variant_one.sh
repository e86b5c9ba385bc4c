#!/bin/bash
# A variant of the library in which ONE source file is rebuilt with extra -D flags, the other objects taken from the regular build:
# tools/lib_variant_one.sh <name> <file without .hip> [-D...] -> afft_amd/lib/libafft_hip_<name>.so   (compare with tools/pp_ab.py)
set -e
cd "$(dirname "$0")/../afft_amd/csrc"
name=$1; file=$2; shift 2
mkdir -p build_var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 "$@" -c $file.hip -o build_var/${file}_$name.o
objs=""
for f in gemm gemm_pp gemm_bd norm attention attention_mfma loss elementwise sublayer; do
  if [ $f = $file ]; then objs="$objs build_var/${file}_$name.o"; else objs="$objs build/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libafft_hip_$name.so $objs

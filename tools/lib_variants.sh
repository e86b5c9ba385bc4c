#!/bin/bash
# A/B of whole-library variants on the GEMM shapes: tools/lib_variants.sh name1 name2 ... ("" = the default library)
for v in "$@"; do
  echo "== $v"
  lib=afft_amd/lib/libafft_hip_$v.so; [ "$v" = "default" ] && lib=afft_amd/lib/libafft_hip.so
  AFFT_LIB=$lib VARIANTS=3 timeout 200 python tools/gemm_bench.py 2>&1 | grep "5120\|8192   8192   8192" | grep -v "2048   2048   5120"
done

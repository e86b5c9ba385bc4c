#!/bin/bash
# A/B of experimental builds of the four-wave GEMM: tools/w4_variants.sh name1 name2 ... (libs afft_amd/lib/libafft_hip_<name>.so)
for v in "$@"; do
  echo "== $v"
  AFFT_LIB=afft_amd/lib/libafft_hip_$v.so VARIANTS=${VARIANTS:-5} timeout 200 python tools/gemm_bench.py 2>&1 | grep "5120\|8192   8192   8192" | grep -v "2048   2048   5120"
done

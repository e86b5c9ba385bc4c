#!/bin/bash
# Diagnostic timing of the general ping-pong GEMM with parts of its loop removed (wrong results).  Build the libraries first:
#   for d in 1 2 4 3 6; do tools/lib_variant_one.sh d$d gemm_pp -DAFFT_DIAG_BUILD -DAFFT_PP_DIAG=$d -DAFFT_PP2=0; done
for d in 1 2 4 3 6; do
  echo "DIAG $d (1 = no fragment reads, 2 = no LDS-DMA in loop, 4 = no MFMA)"
  AFFT_LIB=afft_amd/lib/libafft_hip_d$d.so timeout 100 python tools/gemm_one.py nt 8192 8192 8192 3 20
  AFFT_LIB=afft_amd/lib/libafft_hip_d$d.so timeout 100 python tools/gemm_one.py nt 5120 6144 2048 3 20
done

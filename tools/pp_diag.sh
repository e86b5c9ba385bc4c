#!/bin/bash
# Diagnostic timing of the ping-pong GEMM with parts of its loop removed (libs built with -DAFFT_PP_DIAG=n; wrong results).
for d in 1 2 4 3 6; do
  echo "DIAG $d (1 = no fragment reads, 2 = no LDS-DMA in loop, 4 = no MFMA)"
  AFFT_LIB=afft_amd/lib/libafft_hip_d$d.so timeout 100 python tools/gemm_one.py nt 8192 8192 8192 3 20
  AFFT_LIB=afft_amd/lib/libafft_hip_d$d.so timeout 100 python tools/gemm_one.py nt 5120 6144 2048 3 20
done

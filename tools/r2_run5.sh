#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
SHAPESET=ek100 VARIANTS=1,3,10,12,14 BLAS=1 timeout 600 python tools/gemm_bench.py > gpurun_out/r2_gemm_ek100.txt 2>&1
cat gpurun_out/r2_gemm_ek100.txt

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "test_gemm or splitk or sgd_epilogue" 2>&1 | tail -15 > gpurun_out/r3_sk_tests.txt
tail -5 gpurun_out/r3_sk_tests.txt
VARIANTS=30,3,33,1 BLAS=1 timeout 600 python tools/gemm_bench.py > gpurun_out/r3_sk_bench.txt 2>&1
cat gpurun_out/r3_sk_bench.txt

#!/bin/bash
# scratch: lo8 A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "fp8_lo_pass or fp16" 2>&1 | tail -5 > gpurun_out/lo8_tests.txt
for v in 0 1; do
  AFFT_LO8=$v timeout 300 python bench.py --precision fp16x2 --no-ek100 --no-power --steps 20 --warmup 5 > gpurun_out/lo8_bench_$v.json 2> gpurun_out/lo8_bench_$v.err
done
AFFT_LO8=1 timeout 900 python -m pytest tests/test_model_gpu.py -q -x -k "fp16x2" 2>&1 | tail -8 > gpurun_out/lo8_model_tests.txt

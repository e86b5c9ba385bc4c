#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "gemm" 2>&1 | tail -3
SHAPESET=ek100 VARIANTS=0,1,3 timeout 600 python tools/gemm_bench.py > gpurun_out/r2_gemm_ek100_auto.txt 2>&1
VARIANTS=0,1,3 timeout 600 python tools/gemm_bench.py > gpurun_out/r2_gemm_cfg2_auto.txt 2>&1
grep -v amdgpu gpurun_out/r2_gemm_ek100_auto.txt; grep -v amdgpu gpurun_out/r2_gemm_cfg2_auto.txt
B="--steps 30 --warmup 8 --no-cpu-baseline --no-parity-mode --no-roofline"
for cfg in cfg2 ek100 cfg4 cfg5 cfg1; do
  extra=""; [ $cfg = cfg1 ] && extra="--batch 4"
  timeout 300 python bench.py --config $cfg $B $extra > gpurun_out/r2_b6_$cfg.log 2>&1
  echo "$cfg $(grep -o '"value": [0-9.]*, "unit": "clips/s".\{0,80\}' gpurun_out/r2_b6_$cfg.log | cut -c1-120)"
done

#!/bin/bash
# round 2, GPU call 1: full GPU test-suite + the wgrad CU-cap sweep on the bench workload
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -40 > gpurun_out/r2_t1.log
for w in 0 96 128 64 0; do
  AFFT_WGRAD_WGS=$w timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_b1_wgs$w.log 2>&1
done
tail -5 gpurun_out/r2_t1.log
grep -h -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/r2_b1_wgs*.log | paste - - 

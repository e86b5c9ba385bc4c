#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
B="--steps 30 --warmup 8 --no-cpu-baseline --no-parity-mode --no-roofline"
for w in 0 240 224 192 0; do
  timeout 300 python bench.py $B --wgrad-wgs $w > gpurun_out/r2_b7_wgs$w.log 2>&1
  echo "cfg2 wgs=$w $(grep -o '"value": [0-9.]*, "unit": "clips/s".\{0,80\}' gpurun_out/r2_b7_wgs$w.log | cut -c1-120)"
done
for b in 128 256 512; do
  AFFT_SGD_BLOCKS=$b timeout 300 python bench.py $B > gpurun_out/r2_b7_sgd$b.log 2>&1
  echo "cfg2 sgd_blocks=$b $(grep -o '"value": [0-9.]*, "unit": "clips/s".\{0,80\}' gpurun_out/r2_b7_sgd$b.log | cut -c1-120)"
done
for m in 8 16 64; do
  timeout 300 python bench.py $B --bucket-melems $m > gpurun_out/r2_b7_bucket$m.log 2>&1
  echo "cfg2 bucket=$m $(grep -o '"value": [0-9.]*, "unit": "clips/s".\{0,80\}' gpurun_out/r2_b7_bucket$m.log | cut -c1-120)"
done

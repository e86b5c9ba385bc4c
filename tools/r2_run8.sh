#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > gpurun_out/r2_t8.log; tail -3 gpurun_out/r2_t8.log
B="--steps 30 --warmup 8 --no-cpu-baseline --no-parity-mode"
for cfg in cfg2 ek100 cfg4; do
  for f in 1 0; do
    AFFT_FUSED_SGD=$f timeout 300 python bench.py --config $cfg $B > gpurun_out/r2_b8_${cfg}_fused$f.log 2>&1
    echo "$cfg fused=$f $(grep -o '"value": [0-9.]*, "unit": "clips/s".\{0,80\}' gpurun_out/r2_b8_${cfg}_fused$f.log | cut -c1-120)"
  done
done
grep -o '"roofline".*' gpurun_out/r2_b8_cfg2_fused1.log | cut -c1-1800

#!/bin/bash
# round 2, GPU call 3: GPU test-suite with the composite entry points, then composite on / off and the one-stream graph on
# the host-bound configurations
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r2_t3.log
tail -4 gpurun_out/r2_t3.log
B="--steps 30 --warmup 8 --no-cpu-baseline --no-parity-mode --no-roofline"
for cfg in ek100 cfg4 cfg2; do
  for comp in 1 0; do
    AFFT_COMPOSITE=$comp timeout 300 python bench.py --config $cfg $B > gpurun_out/r2_b3_${cfg}_comp$comp.log 2>&1
    echo "$cfg composite=$comp $(grep -o '"value": [0-9.]*, "unit": "clips/s".\{0,80\}' gpurun_out/r2_b3_${cfg}_comp$comp.log | cut -c1-120)"
  done
done
for cfg in ek100 cfg4; do
  timeout 300 python bench.py --config $cfg $B --graph single > gpurun_out/r2_b3_${cfg}_graph1.log 2>&1
  echo "$cfg graph-single $(grep -o '"value": [0-9.]*, "unit": "clips/s".\{0,80\}' gpurun_out/r2_b3_${cfg}_graph1.log | cut -c1-120)"
  timeout 300 python bench.py --config $cfg $B --graph on > gpurun_out/r2_b3_${cfg}_graph3.log 2>&1
  echo "$cfg graph-3streams $(grep -o '"value": [0-9.]*, "unit": "clips/s".\{0,80\}' gpurun_out/r2_b3_${cfg}_graph3.log | cut -c1-120)"
done
timeout 300 python tools/host_profile.py ek100 64 > gpurun_out/r2_hostprof_ek100.txt 2>&1
AFFT_COMPOSITE=0 timeout 300 python tools/host_profile.py ek100 64 > gpurun_out/r2_hostprof_ek100_comp0.txt 2>&1
head -30 gpurun_out/r2_hostprof_ek100.txt

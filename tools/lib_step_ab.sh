#!/bin/bash
# Step time of the bench workload with two builds of the library, alternating processes on ONE box (processes on one box repeat to ~0.2 %).
# usage (GPU box): bash tools/lib_step_ab.sh <other lib path> [rounds] [extra bench flags]
other=$1; rounds=${2:-3}; shift 2
flags="--steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-parity-mode --no-reference-loop --no-power --no-ek100 --no-small-batch --no-separate-update $*"
for r in $(seq 1 $rounds); do
  for lib in product $other; do
    if [ $lib = product ]; then unset AFFT_LIB; else export AFFT_LIB=$lib; fi
    out=$(timeout 600 python bench.py $flags 2>/dev/null | grep '^{' | tail -1)
    echo "$lib $(python -c "import json,sys; d=json.loads(sys.argv[1]); print(d['ms_per_step'], d.get('fwd_p50_ms'), d.get('final_loss'))" "$out")"
  done
done

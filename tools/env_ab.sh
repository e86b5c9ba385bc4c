#!/bin/bash
# Alternating bench.py processes on one box under two (or more) environment settings: ms/step, forward p50, final loss per process.
# usage (GPU box): N=4 bash tools/env_ab.sh "AFFT_BD_MODE=1" "AFFT_BD_MODE=0"   -> gpurun_out/r06_env_ab.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
F="--no-cpu-baseline --no-parity-mode --no-reference-loop --no-power --no-ek100 --no-small-batch --no-roofline --steps 30 --warmup 8 ${BENCH_ARGS}"
for i in $(seq ${N:-4}); do
  for setting in "$@"; do
    r=$(env $setting timeout 300 python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'fwd_p50', d['fwd_p50_ms'], 'loss', d['final_loss'])")
    echo "$setting run $i: $r" | tee -a gpurun_out/r06_env_ab.txt
  done
done

#!/bin/bash
# Small-batch / small-width regimes: the reference's own widths (ek100) and per-GPU batch (16) against the bench
# workload, eager vs hipGraph replay, and the size threshold below which a weight's update is left to the per-bucket kernel instead of
# its weight-gradient epilogue (AFFT_FUSE_MIN_ELEMS).  usage (GPU box): bash tools/small_batch_sweep.sh  -> gpurun_out/r06_small_batch.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
F="--no-cpu-baseline --no-parity-mode --no-reference-loop --no-power --no-ek100 --no-roofline --steps 30 --warmup 8"
out=gpurun_out/r06_small_batch.txt; : > $out
run() { # cfg batch graph min_elems
  r=$(AFFT_FUSE_MIN_ELEMS=$4 timeout 300 python bench.py --config $1 --batch $2 --graph $3 $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['config']['step_launch'], d['optimizer_path'])")
  echo "$1 B=$2 graph=$3 fuse_min_elems=$4: $r" | tee -a $out
}
for cfg in cfg2 ek100; do for b in 64 16; do
  for t in 0 4500000 13000000 1000000000; do run $cfg $b off $t; done
  run $cfg $b on 1000000000
  run $cfg $b on 13000000
done; done

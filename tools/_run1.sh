#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( bash tools/cfg_sweep.sh --no-parity-mode --no-reference-loop --no-power --no-ek100
  for b in 16 128; do echo -n "B=$b "; CFGS=cfg2 bash tools/cfg_sweep.sh --batch $b --no-parity-mode --no-reference-loop --no-power --no-ek100; done
  echo -n "fp16x2 "; CFGS="cfg2 ek100 cfg5" bash tools/cfg_sweep.sh --precision fp16x2 --no-parity-mode --no-reference-loop --no-power --no-ek100 ) > gpurun_out/r05_configs.txt 2>&1
cat gpurun_out/r05_configs.txt

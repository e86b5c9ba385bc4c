#!/bin/bash
# step time vs the grid cap of the SGD kernel (one box): tools/sgd_blocks.sh
cd "$(dirname "$0")/.."
for b in ${SWEEP:-4096 1024 512 256 128 4096 512}; do
  echo -n "AFFT_SGD_BLOCKS=$b  "
  AFFT_SGD_BLOCKS=$b timeout 300 python bench.py --steps 30 --warmup 8 --no-roofline --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip()); print(r['value'], 'clips/s', r['ms_per_step'], 'ms')"
done

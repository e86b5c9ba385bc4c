#!/bin/bash
# eager vs captured (hipGraph) step, same box: tools/graph_ab.sh
cd "$(dirname "$0")/.."
for cfg in cfg2 ek100 cfg4 cfg2_cm; do
  for g in off on; do
    echo -n "$cfg graph=$g  "
    timeout 300 python bench.py --config $cfg --graph $g --steps 30 --warmup 8 --no-roofline --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
l = sys.stdin.read().strip()
try:
    r = json.loads(l); print(r['value'], 'clips/s', r['ms_per_step'], 'ms  loss', r['final_loss'])
except Exception: print('ERR', l[-400:])"
  done
done

#!/bin/bash
# throughput of every bench configuration (one box): tools/cfg_sweep.sh [extra bench args]
cd "$(dirname "$0")/.."
for cfg in ${CFGS:-cfg2 ek100 cfg4 cfg5 cfg2_cm cfg2_tsa}; do
  echo -n "$cfg  "
  timeout 300 python bench.py --config $cfg --steps 30 --warmup 8 --no-roofline --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "
import sys, json
l = sys.stdin.read().strip()
try:
    r = json.loads(l); print(r['value'], 'clips/s', r['ms_per_step'], 'ms  fwd p50', r.get('fwd_p50_ms'))
except Exception: print('ERR', l[-300:])"
done

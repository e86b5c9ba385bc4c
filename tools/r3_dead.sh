#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -x -q -k "last_block or reference_golden or full_size_matches or full_width" 2>&1 | grep -E "passed|failed|^FAILED|Error" | tail -5
for cfg in cfg2 ek100 cfg5; do for fr in "" "--full-rows"; do
  echo "$cfg $fr: $(python bench.py --config $cfg $fr --steps 20 --warmup 5 --no-cpu-baseline --no-parity-mode --no-roofline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["final_loss"], d["fwd_p50_ms"], d["mfma_frac_whole_step"])')"
done; done

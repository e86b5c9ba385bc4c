#!/bin/bash
# round 2, GPU call 4: timelines of the reference-faithful EK100 configuration and cfg4 (where does the time go?) + new tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -x -k "marginalize or drop_path or out_of_range" 2>&1 | tail -4
bash tools/prof_timeline.sh r2_ek100 --no-parity-mode --config ek100
bash tools/prof_timeline.sh r2_cfg4 --no-parity-mode --config cfg4
for t in r2_ek100 r2_cfg4; do
  python - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/prof_$t/**/*kernel_stats.csv", recursive=True)
print("$t stats files", f[:1])
PY
  find gpurun_out/prof_$t -name "*kernel_trace.csv" -delete; find gpurun_out/prof_$t -name "*.db" -delete
done
timeout 300 python bench.py --config ek100 --steps 30 --warmup 8 --no-cpu-baseline --no-parity-mode --no-roofline --no-optimizer > gpurun_out/r2_b4_ek100_noopt.log 2>&1
grep -o '"value": [0-9.]*, "unit": "clips/s".\{0,80\}' gpurun_out/r2_b4_ek100_noopt.log | cut -c1-120
timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity-mode --no-roofline --no-optimizer > gpurun_out/r2_b4_cfg2_noopt.log 2>&1
grep -o '"value": [0-9.]*, "unit": "clips/s".\{0,80\}' gpurun_out/r2_b4_cfg2_noopt.log | cut -c1-120

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r3_04_tests.txt; cat gpurun_out/r3_04_tests.txt
for cap in 0 160 192 224 240; do
  echo "cap $cap: $(python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-mode --no-roofline --wgrad-wgs $cap 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done > gpurun_out/r3_cap_sweep.txt 2>&1
AFFT_OVERLAP_WGRAD=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-mode --no-roofline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("serial", d["value"], d["ms_per_step"])' >> gpurun_out/r3_cap_sweep.txt
cat gpurun_out/r3_cap_sweep.txt

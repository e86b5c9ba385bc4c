#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_model_gpu.py -q -x -k "non_finite or fused or reproducible or dropin or reference_loop or captured" > gpurun_out/t.log 2>&1; grep -E "passed|failed|Fatal|^E  " gpurun_out/t.log | tail -8
for i in 1 2; do timeout 300 python bench.py --no-ek100 --no-power --no-cpu-baseline --no-parity-mode --no-reference-loop --no-roofline --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16', d['value'], d['ms_per_step'])"; done

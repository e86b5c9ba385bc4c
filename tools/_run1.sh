#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do
timeout 300 python bench.py --no-ek100 --no-power --no-cpu-baseline --no-parity-mode --no-reference-loop --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 new', d['value'], d['ms_per_step'])"
AFFT_LIB=$GRAFT_REPO_ROOT/tools/experiments/libafft_prev.so timeout 300 python bench.py --no-ek100 --no-power --no-cpu-baseline --no-parity-mode --no-reference-loop --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 prev lib', d['value'], d['ms_per_step'])"
done

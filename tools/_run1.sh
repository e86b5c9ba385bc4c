#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_model_gpu.py -q -x -k "token_rows" 2>&1 | tail -3
for i in 1 2 3; do for t in 0 1; do AFFT_ATTN_TAKE=$t timeout 300 python bench.py --no-ek100 --no-power --no-cpu-baseline --no-parity-mode --no-reference-loop --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 take=$t', d['value'], d['ms_per_step'])"; done; done
for t in 0 1; do AFFT_ATTN_TAKE=$t timeout 300 python bench.py --precision fp16x2 --no-ek100 --no-power --no-cpu-baseline --no-parity-mode --no-reference-loop --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp16x2 take=$t', d['value'], d['ms_per_step'])"; done

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention" 2>&1 | tail -2
for i in 1 2; do for v in "" _pl160 _pl32; do
AFFT_LIB=$GRAFT_REPO_ROOT/afft_amd/lib/libafft_hip$v.so timeout 300 python bench.py --precision fp16x2 --no-ek100 --no-power --no-cpu-baseline --no-parity-mode --no-reference-loop --no-roofline --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp16x2 lib$v', d['value'], d['ms_per_step'], d.get('fwd_p50_ms'))"; done; done

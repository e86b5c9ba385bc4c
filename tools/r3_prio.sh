#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
V=$GRAFT_REPO_ROOT/afft_amd/lib/libafft_hip_prio0.so
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity-mode --no-roofline > /dev/null 2>&1
for rep in 1 2 3; do for cfg in cfg2 ek100 cfg4; do
  echo "$cfg default: $(python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-parity-mode --no-roofline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["fwd_p50_ms"])')"
  echo "$cfg prio0:   $(AFFT_LIB=$V python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-parity-mode --no-roofline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["fwd_p50_ms"])')"
done; done
VARIANTS=30 python tools/gemm_bench.py 2>&1 | grep -v amdgpu | head -12
AFFT_LIB=$V VARIANTS=30 python tools/gemm_bench.py 2>&1 | grep -v amdgpu | head -12

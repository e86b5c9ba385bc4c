cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05c
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "layernorm or attention" 2>&1 | tail -4
timeout 600 python bench.py --no-cpu-baseline --no-parity-mode --no-reference-loop --no-ek100 --steps 20 --warmup 5 > gpurun_out/r05c/bench.json 2> gpurun_out/r05c/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05c/bench.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ['value','ms_per_step','fwd_p50_ms']})
print(d['hbm_kernels'])
print(d['roofline']['by_k_class'])
PY

cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05b
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" 2>&1 | tail -4
timeout 600 python bench.py --precision fp16x2 --no-cpu-baseline --no-parity-mode --no-reference-loop --no-ek100 --steps 20 --warmup 5 > gpurun_out/r05b/bench_f16x2.json 2> gpurun_out/r05b/bench_f16x2.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05b/bench_f16x2.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ['value','ms_per_step','fwd_p50_ms','fwd_p50_roofline']})
for k,v in d['roofline']['by_kernel'].items(): print(k, v)
print(d['roofline'].get('by_k_class'))
PY
( time timeout 900 python bench.py ) > gpurun_out/r05b/bench.json 2> gpurun_out/r05b/bench.err
tail -c 3000 gpurun_out/r05b/bench.json

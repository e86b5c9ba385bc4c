cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05c
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" 2>&1 | tail -12
for v in 0 1; do
AFFT_ATTN_BWD_STAGED=$v timeout 600 python bench.py --no-cpu-baseline --no-parity-mode --no-reference-loop --no-ek100 --no-power --steps 20 --warmup 5 > gpurun_out/r05c/bench_at$v.json 2> gpurun_out/r05c/bench.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r05c/bench_at$v.json').read().strip().splitlines()[-1])
print('staged=$v', {k:d.get(k) for k in ['value','ms_per_step','fwd_p50_ms']}, d['hbm_kernels']['attn_bwd'])
PY
done

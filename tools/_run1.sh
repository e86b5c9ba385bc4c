cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05e
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "lds_ring or test_gemm" 2>&1 | tail -3
for lib in one two; do
  if [ $lib = two ]; then export AFFT_LIB=$PWD/afft_amd/lib/libafft_hip_2bar.so; else unset AFFT_LIB; fi
  echo "== $lib barrier(s) per phase" | tee -a gpurun_out/r05e/gemm_ab.txt
  VARIANTS=3 timeout 300 python tools/gemm_bench.py 2>&1 | grep -v "^$" | head -13 | tee -a gpurun_out/r05e/gemm_ab.txt
done
for rep in 1 2; do for lib in one two; do
  if [ $lib = two ]; then export AFFT_LIB=$PWD/afft_amd/lib/libafft_hip_2bar.so; else unset AFFT_LIB; fi
  timeout 600 python bench.py --no-cpu-baseline --no-parity-mode --no-reference-loop --no-ek100 --no-power --steps 30 --warmup 8 > gpurun_out/r05e/bench_$lib$rep.json 2>/dev/null
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r05e/bench_$lib$rep.json').read().splitlines() if l.startswith('{')][-1])
print('$lib rep $rep', {k:d.get(k) for k in ['value','ms_per_step','fwd_p50_ms']}, 'dom frac', d['roofline']['frac'], 'alone', d['roofline']['alone']['frac'])
PY
done; done 2>&1 | tee -a gpurun_out/r05e/gemm_ab.txt

cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
F="--no-cpu-baseline --no-parity-mode --no-reference-loop --no-power --no-ek100 --no-roofline --steps 30 --warmup 8"
for i in $(seq ${N:-6}); do
  for lib in ${LIBS:-product g2off}; do
    if [ $lib = product ]; then unset AFFT_LIB; else export AFFT_LIB=$PWD/afft_amd/lib/libafft_hip_$lib.so; fi
    r=$(timeout 300 python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['final_loss'], 'fwd_p50', d['fwd_p50_ms'])")
    echo "$lib run $i: $r" | tee -a gpurun_out/r06_nan_hunt.txt
  done
done

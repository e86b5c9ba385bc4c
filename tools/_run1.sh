cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05f
for d in 0 1 2 3 7; do
  if [ $d = 0 ]; then unset AFFT_LIB; else export AFFT_LIB=$PWD/afft_amd/lib/libafft_hip_q4d$d.so; fi
  python - <<PY 2>&1 | tee -a gpurun_out/r05f/q4_diag.txt
import sys, os
sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import gemm_bench as GB
r = GB.bench("nt", 8192, 8192, 8192, 11)
print("diag $d: q4 8192^3 %.4f ms %.1f TFLOP/s" % r)
PY
done
unset AFFT_LIB
python tools/q4_power.py 2>&1 | tail -5 | tee -a gpurun_out/r05f/q4_diag.txt

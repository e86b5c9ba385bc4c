#!/bin/bash
# which hipBLASLt kernels does torch.mm pick for the path's shapes?  (names encode macro-tile, workgroup, LDS options)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_blas
VARIANTS=3 BLAS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_blas -o bl -- python tools/gemm_bench.py > gpurun_out/prof_blas.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_blas/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "Cijk" in n or "gemm" in n.lower():
        print(r["Calls"], r["AverageNs"], n[:400])
PY

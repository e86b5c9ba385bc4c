#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/st.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from tools.gemm_bench import bench
name = os.environ.get("AFFT_LIB", "default").split("libafft_hip")[-1]
for lay, M, N, K in (("nt", 5120, 8192, 2048), ("nt", 5120, 2048, 8192), ("nt", 8192, 8192, 8192), ("nt", 1024, 6144, 2048), ("nn", 5120, 8192, 2048), ("tn", 2048, 8192, 5120), ("tn", 8192, 8192, 8192)):
    t = min(bench(lay, M, N, K, 30)[0] for _ in range(3))
    tiles = ((M + 255) // 256) * ((N + 255) // 256); rounds = (tiles + 255) // 256
    print(name, lay, M, N, K, "%.1f us %.0f TF  | %.2f us per K-tile and round" % (t * 1e3, 2.0 * M * N * K / t / 1e9, t * 1e3 / (rounds * K / 64)))
PY
python /tmp/st.py 2>&1 | grep -v amdgpu.ids
AFFT_LIB=$GRAFT_REPO_ROOT/afft_amd/lib/libafft_hip_sametile.so python /tmp/st.py 2>&1 | grep -v amdgpu.ids

#!/bin/bash
# A/B of four-wave kernel variants (tools/w4_variant.sh) on NT shapes: us per launch, variant 5 (LDS-DMA staging); diagnostic builds give wrong results
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cat > /tmp/w4ab.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from tools.gemm_bench import bench
name = os.environ.get("AFFT_LIB", "default").split("libafft_hip")[-1]
r = []
for lay, M, N, K in (("nt", 8192, 8192, 8192), ("nt", 5120, 8192, 2048), ("nt", 5120, 2048, 8192), ("tn", 2048, 8192, 5120), ("nn", 5120, 8192, 2048)):
    t5 = min(bench(lay, M, N, K, 5)[0] for _ in range(3)); t3 = min(bench(lay, M, N, K, 30)[0] for _ in range(2))
    r.append("%s %dx%dx%d w4 %.1f pp %.1f" % (lay, M, N, K, t5 * 1e3, t3 * 1e3))
print(name, " | ".join(r))
PY
for v in "$@"; do AFFT_LIB=$GRAFT_REPO_ROOT/afft_amd/lib/libafft_hip_$v.so python /tmp/w4ab.py 2>&1 | grep -v amdgpu.ids; done

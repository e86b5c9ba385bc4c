#!/bin/bash
# 128x128 kernel: split-K off / 2 / 4 on the small-grid shapes of the path with the round-3 (16-byte) hand-over
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cat > /tmp/sk.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from tools.gemm_bench import bench
shapes = [("nt", 1024, 2048, 2048), ("nn", 1024, 2048, 2048), ("nt", 1024, 2048, 6144), ("nt", 1024, 2048, 8192), ("nn", 1024, 2048, 8192),
          ("nt", 1024, 6144, 2048), ("nn", 1024, 6144, 2048), ("nt", 1024, 8192, 2048), ("nn", 1024, 8192, 2048),
          ("tn", 2048, 2048, 1024), ("tn", 2048, 2048, 5120), ("tn", 1024, 4096, 5120), ("tn", 3072, 1024, 5120), ("tn", 1024, 1024, 5120),
          ("nt", 1088, 3840, 2048), ("nt", 1088, 2048, 3840), ("tn", 3840, 2048, 1088), ("nt", 5120, 1024, 4096), ("nn", 5120, 1024, 4096), ("nn", 5120, 1024, 3072)]
print(f"{'layout':6} {'M':>6} {'N':>6} {'K':>6} | auto us | no split | split 2 | split 4")
for lay, M, N, K in shapes:
    r = [min(bench(lay, M, N, K, v)[0] for _ in range(2)) * 1e3 for v in (1, 10, 12, 14)]
    print(f"{lay:6} {M:6d} {N:6d} {K:6d} | {r[0]:7.1f} | {r[1]:8.1f} | {r[2]:7.1f} | {r[3]:7.1f}")
PY
python /tmp/sk.py 2>&1 | grep -v amdgpu

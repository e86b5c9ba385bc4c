#!/bin/bash
# hand-off cost of the stream-K kernel, by diagnostic builds (tools/pp_variant.sh)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cat > /tmp/skd.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from tools.gemm_bench import bench
r = []
for K in (1024, 2048, 8192):
    r.append("%d: plain %.1f sk %.1f" % (K, bench("nt", 5120, 2048, K, 30)[0] * 1e3, bench("nt", 5120, 2048, K, 33)[0] * 1e3))
for K in (512, 2048):
    r.append("N8192 %d: plain %.1f sk %.1f" % (K, bench("nt", 5120, 8192, K, 30)[0] * 1e3, bench("nt", 5120, 8192, K, 33)[0] * 1e3))
print(os.environ.get("AFFT_LIB", "default").split("_")[-1], " | ".join(r))
PY
for v in "" aux16 nostore noload noxchg; do
  if [ -z "$v" ]; then python /tmp/skd.py; else AFFT_LIB=$GRAFT_REPO_ROOT/afft_amd/lib/libafft_hip_$v.so python /tmp/skd.py; fi
done 2>&1 | grep -v amdgpu.ids

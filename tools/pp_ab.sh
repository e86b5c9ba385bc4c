#!/bin/bash
# A/B of gemm_pp.hip variants (tools/pp_variant.sh, NT layout only) on the NT shapes of the path + a correctness check
# usage: tools/pp_ab.sh <variant> ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cat > /tmp/ppab.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from tools.gemm_bench import bench
from afft_amd import ops
name = os.environ.get("AFFT_LIB", "default").split("libafft_hip")[-1]
# correctness against torch on one shape
a = torch.randn(1024, 2048, device="cuda").to(torch.bfloat16); b = torch.randn(768, 2048, device="cuda").to(torch.bfloat16)
from afft_amd import _lib
_lib.check(_lib.lib().afft_set_gemm_variant(3))
out = torch.empty(1024, 768, dtype=torch.bfloat16, device="cuda")
ops.gemm(a, b, out, b_t=True)
ref = (a.float() @ b.float().t())
err = float((out.float() - ref).norm() / ref.norm())
r = []
for M, N, K in ((5120, 8192, 2048), (5120, 2048, 8192), (5120, 6144, 2048), (5120, 2048, 2048), (8192, 8192, 8192)):
    t = min(bench("nt", M, N, K, 30)[0] for _ in range(3))
    r.append("%dx%dx%d %.1f us %.0f TF" % (M, N, K, t * 1e3, 2.0 * M * N * K / t / 1e9))
print(name, "err %.1e |" % err, " | ".join(r))
PY
python /tmp/ppab.py 2>&1 | grep -v amdgpu.ids
for v in "$@"; do AFFT_LIB=$GRAFT_REPO_ROOT/afft_amd/lib/libafft_hip_$v.so python /tmp/ppab.py 2>&1 | grep -v amdgpu.ids; done

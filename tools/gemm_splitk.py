"""Split-K (last-arriver reduction) vs plain on the small-grid shapes, fp32 output with bias+residual epilogue."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afft_amd import _lib, ops
dev = "cuda:0"
def run(lay, M, N, K, split, iters=20):
    _lib.check(_lib.lib().afft_set_gemm_splitk(split))
    a = torch.randn(M, K).to(torch.bfloat16).to(dev)
    b = (torch.randn(N, K) if lay == "nt" else torch.randn(K, N)).to(torch.bfloat16).to(dev)
    res = torch.randn(M, N, device=dev); bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    kw = dict(b_t=True) if lay == "nt" else {}
    for _ in range(3):
        ops.gemm(a, b, out, bias=bias, residual=res, **kw)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        ops.gemm(a, b, out, bias=bias, residual=res, **kw)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    return ms * 1e3
for lay, M, N, K in [("nn", 1024, 2048, 2048), ("nn", 1024, 2048, 8192), ("nt", 1024, 2048, 2048), ("nt", 1024, 2048, 6144),
                     ("nt", 1024, 2048, 8192), ("nt", 1024, 6144, 2048), ("nt", 1024, 8192, 2048), ("nt", 1024, 3840, 2048)]:
    print(lay, M, N, K, "plain us %.1f  auto %.1f  x2 %.1f  x4 %.1f" % tuple(run(lay, M, N, K, m) for m in (0, 1, 2, 4)))
_lib.check(_lib.lib().afft_set_gemm_splitk(1))

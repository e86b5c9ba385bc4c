"""Does a non-power-of-two leading dimension change the GEMM rate (L2 channel / TCP set conflicts)?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afft_amd import _lib, ops

def run(M, N, K, pad, variant, iters=20):
    _lib.check(_lib.lib().afft_set_gemm_variant(variant))
    dev = "cuda:0"
    a = torch.randn(M, K + pad).to(torch.bfloat16).to(dev)[:, :K]
    b = torch.randn(N, K + pad).to(torch.bfloat16).to(dev)[:, :K]
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for _ in range(3):
        ops.gemm(a, b, out, b_t=True)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        ops.gemm(a, b, out, b_t=True)
    e.record(); torch.cuda.synchronize()
    return 2.0 * M * N * K / (s.elapsed_time(e) / iters) / 1e9

for (M, N, K) in [(5120, 8192, 2048), (8192, 8192, 8192), (1024, 2048, 2048)]:
    for v in (1, 3):
        print(M, N, K, "variant", v, " | ".join(f"pad {p}: {run(M, N, K, p, v):7.1f}" for p in (0, 64, 192, 8)))

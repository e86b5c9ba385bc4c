"""Numerics + timing of the four-quadrant GEMM kernel (variant 11) against the ping-pong kernel (variant 3) and the vendor library."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afft_amd import _lib, ops
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import gemm_bench as GB

dev = "cuda:0"
g = torch.Generator().manual_seed(0)
for (M, N, K) in [(256, 256, 64), (512, 768, 256), (300, 520, 192), (5120, 2048, 2048), (1000, 3806 // 2 * 2, 2048)]:
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    b = torch.randn(N, K, generator=g).to(torch.bfloat16).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    outs = {}
    for v in (3, 11):
        _lib.check(_lib.lib().afft_set_gemm_variant(v))
        o = torch.zeros(M, N, device=dev)
        ops.gemm(a, b, o, b_t=True, bias=bias, act=1)
        torch.cuda.synchronize()
        outs[v] = o
    _lib.check(_lib.lib().afft_set_gemm_variant(0))
    ref = torch.nn.functional.gelu(a.float().double() @ b.float().double().t() + bias.double()).float()
    e3 = float((outs[3] - ref).norm() / ref.norm()); e11 = float((outs[11] - ref).norm() / ref.norm())
    print(f"{M}x{N}x{K}: rel err pp {e3:.2e} q4 {e11:.2e} max|q4-pp| {float((outs[11]-outs[3]).abs().max()):.2e}")
print("layout M N K | pp ms TF | q4 ms TF | blas ms TF")
for (M, N, K) in [(5120, 6144, 2048), (5120, 2048, 2048), (5120, 8192, 2048), (5120, 2048, 8192), (8192, 8192, 8192), (1024, 8192, 2048)]:
    r = [GB.bench("nt", M, N, K, v) for v in (3, 11)] + [GB.bench_blas("nt", M, N, K)]
    print(f"nt {M} {N} {K} | " + " | ".join(f"{x[0]:.4f} {x[1]:7.1f}" for x in r))

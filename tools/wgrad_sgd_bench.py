"""Weight-gradient GEMMs with the Nesterov update in their epilogue (the step's dominant kernel family): 128x128 vs 256x256 tiles
per shape, cold operands (each launch its own weight / momentum buffers: 18 B per element of optimizer traffic is the point).
usage: python tools/wgrad_sgd_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afft_amd import _lib, ops  # noqa: E402

dev = "cuda:0"


def run(M, N, K, variant, splitk=1, fused=True, nbuf=6, iters=18):
    _lib.check(_lib.lib().afft_set_gemm_variant(variant))
    _lib.check(_lib.lib().afft_set_gemm_splitk(splitk))
    a = torch.randn(K, M, device=dev).to(torch.bfloat16)
    b = torch.randn(K, N, device=dev).to(torch.bfloat16)
    bufs = []
    for _ in range(nbuf):
        p, m, p16 = torch.randn(M, N, device=dev), torch.zeros(M, N, device=dev), torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
        d = _lib.SgdFused()
        d.p, d.buf, d.p_bf16, d.lr, d.mom, d.wd, d.gscale, d.first_step = p.data_ptr(), m.data_ptr(), p16.data_ptr(), 1e-3, 0.9, 1e-6, 1.0, 0
        bufs.append((p, m, p16, d, torch.empty(M, N, device=dev)))
    def once(i):
        p, m, p16, d, g = bufs[i % nbuf]
        ops.gemm(a, b, g, a_t=True, sgd=d if fused else None)
    for i in range(nbuf):
        once(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters):
        once(i)
    e.record()
    torch.cuda.synchronize()
    _lib.check(_lib.lib().afft_set_gemm_variant(0))
    _lib.check(_lib.lib().afft_set_gemm_splitk(1))
    return s.elapsed_time(e) / iters * 1e3


if __name__ == "__main__":
    print(f"{'M':>6} {'N':>6} {'K':>6} | auto us | 128x128 us | 128x128 no split | 256x256 us | plain (gradient stored) auto us | optimizer bytes / 5 TB/s us")
    for M, N, K in ((2048, 8192, 1024), (8192, 2048, 1024), (2048, 6144, 1024), (2048, 2048, 1024), (2048, 8192, 5120), (8192, 2048, 5120),
                    (6144, 2048, 5120), (2048, 2048, 5120), (1024, 4096, 5120), (3072, 1024, 5120)):
        r = [run(M, N, K, 0), run(M, N, K, 1), run(M, N, K, 1, splitk=0), run(M, N, K, 3), run(M, N, K, 0, fused=False)]
        print(f"{M:6d} {N:6d} {K:6d} | {r[0]:7.1f} | {r[1]:10.1f} | {r[2]:16.1f} | {r[3]:10.1f} | {r[4]:31.1f} | {M * N * 18 / 5e6:8.1f}")

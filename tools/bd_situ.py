"""Why the B-direct kernel gains less inside the model than alone: the path's fc2 / projection shapes timed one launch at a
time with (a) warm or cold caches (a 1 GiB fill between launches evicts L2 + Infinity Cache) and (b) the bench's plain bf16
output or the model's epilogue (bias + fp32 residual read + fp32 output).  usage: python tools/bd_situ.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afft_amd import _lib, ops  # noqa: E402
from tools.bd_check import pack_b  # noqa: E402

DEV = "cuda:0"
g = torch.Generator().manual_seed(0)
flush = torch.empty(1 << 28, dtype=torch.float32, device=DEV)      # 1 GiB


def one(fn, cold, n=12):
    ts = []
    for _ in range(n):
        if cold:
            flush.fill_(1.0)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for (M, N, K) in [(5120, 2048, 8192), (5120, 2048, 2048), (5120, 8192, 2048)]:
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(DEV)
    b = torch.randn(N, K, generator=g).to(torch.bfloat16).to(DEV)
    bp = pack_b(b)
    bias = torch.randn(N, generator=g).to(DEV)
    res = torch.randn(M, N, generator=g).to(DEV)
    o16 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    o32 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    for v, bb in ((3, b), (10, bp)):
        _lib.check(_lib.lib().afft_set_gemm_variant(v))
        row = []
        for cold in (False, True):
            row.append(one(lambda: ops.gemm(a, bb, o16, b_t=True), cold))
            row.append(one(lambda: ops.gemm(a, bb, o32, b_t=True, bias=bias, residual=res), cold))
        print(f"{M}x{N}x{K} v{v}: warm bf16-out {row[0]:6.1f} | warm bias+res+f32 {row[1]:6.1f} | cold bf16-out {row[2]:6.1f} | cold bias+res+f32 {row[3]:6.1f} us", flush=True)
    _lib.check(_lib.lib().afft_set_gemm_variant(0))
    bt = b.t()
    o32b = torch.empty(M, N, dtype=torch.float32, device=DEV)
    row = [one(lambda: torch.mm(a, bt, out=o16), False), one(lambda: torch.mm(a, bt, out=o16), True)]
    print(f"{M}x{N}x{K} vendor (torch.mm, bf16 out; a yard-stick only): warm {row[0]:6.1f} | cold {row[1]:6.1f} us", flush=True)

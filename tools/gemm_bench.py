"""Times the bf16 MFMA GEMM variants on the shapes of the cfg2 (B=64) step.  Usage: python tools/gemm_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afft_amd import _lib, ops  # noqa: E402


def bench(layout, M, N, K, variant, iters=20):
    """variant: 1 / 3 (tile shape) with automatic split-K; 10 = 128x128 tile without split-K; 12 / 14 = 128x128, split-K 2 / 4 forced;
    30 = 256x256 tile without split-K; 32 / 33 = 256x256, split-K 2 / 3 forced; 100 = the library's own choice when the weight
    comes with a fragment-packed image (NT only: what the model's forward GEMMs get); 110 / 109 = B-direct 160x256 / 256x256 tiles
    on the packed image, forced"""
    dev = "cuda:0"
    packed = variant in (100, 109, 110)
    if packed and layout != "nt":
        variant = 0
        packed = False
    forced = {100: 0, 109: 9, 110: 10}.get(variant)
    _lib.check(_lib.lib().afft_set_gemm_splitk({10: 0, 12: 2, 14: 4, 30: 0, 32: 2, 33: 4}.get(variant, 1)))
    variant = forced if forced is not None else 3 if variant >= 30 else 1 if variant >= 10 else variant      # 0 = the library's own choice
    _lib.check(_lib.lib().afft_set_gemm_variant(variant))
    g = torch.Generator().manual_seed(0)
    if layout == "nt":
        a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev); b = torch.randn(N, K, generator=g).to(torch.bfloat16).to(dev)
        kw = dict(b_t=True)
        if packed:
            pk = torch.empty(N * K, dtype=torch.bfloat16, device=dev)
            ops.pack_weight(b.float(), pk)
            if variant in (9, 10):       # forced: B itself is the packed image
                kw = dict(b_t=True, b_packed=pk)
                b = pk.view(N, K)
            else:
                kw = dict(b_t=True, b_packed=pk)
    elif layout == "nn":
        a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev); b = torch.randn(K, N, generator=g).to(torch.bfloat16).to(dev)
        kw = dict()
    else:  # tn: a [K, M], b [K, N]
        a = torch.randn(K, M, generator=g).to(torch.bfloat16).to(dev); b = torch.randn(K, N, generator=g).to(torch.bfloat16).to(dev)
        kw = dict(a_t=True)
    out = torch.empty(M, N, dtype=torch.bfloat16 if layout != "tn" else torch.float32, device=dev)
    for _ in range(3):
        ops.gemm(a, b, out, **kw)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        ops.gemm(a, b, out, **kw)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    _lib.check(_lib.lib().afft_set_gemm_variant(0))
    return ms, 2.0 * M * N * K / ms / 1e9


def bench_blas(layout, M, N, K, iters=20):
    """The vendor library (torch.mm -> hipBLASLt/rocBLAS) on the same operands: a yardstick, never the product path."""
    dev = "cuda:0"
    if layout == "nt":
        a = torch.randn(M, K, device=dev).to(torch.bfloat16); b = torch.randn(N, K, device=dev).to(torch.bfloat16).t()
    elif layout == "nn":
        a = torch.randn(M, K, device=dev).to(torch.bfloat16); b = torch.randn(K, N, device=dev).to(torch.bfloat16)
    else:
        a = torch.randn(K, M, device=dev).to(torch.bfloat16).t(); b = torch.randn(K, N, device=dev).to(torch.bfloat16)
    for _ in range(3):
        torch.mm(a, b)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        torch.mm(a, b)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    return ms, 2.0 * M * N * K / ms / 1e9


if __name__ == "__main__":
    shapes = [("nt", 5120, 6144, 2048), ("nt", 5120, 2048, 2048), ("nt", 5120, 8192, 2048), ("nt", 5120, 2048, 8192),
              ("nn", 5120, 2048, 6144), ("nn", 5120, 8192, 2048), ("nn", 5120, 2048, 8192),
              ("tn", 6144, 2048, 5120), ("tn", 8192, 2048, 5120), ("tn", 2048, 8192, 5120), ("tn", 2048, 2048, 5120),
              ("nt", 1024, 6144, 2048), ("nt", 1024, 8192, 2048), ("nt", 1024, 2048, 8192), ("tn", 8192, 2048, 1024),
              ("nt", 1024, 2048, 2048), ("tn", 2048, 2048, 1024), ("tn", 6144, 2048, 1024), ("nt", 1024, 3840, 2048),
              ("nt", 1088, 3840, 2048), ("nt", 8192, 8192, 8192)]
    if os.environ.get("SHAPESET") == "ek100":    # the reference's own EK100 widths (d = 1024, D = 2048), B = 64
        shapes = [("nt", 5120, 3072, 1024), ("nt", 5120, 1024, 1024), ("nt", 5120, 4096, 1024), ("nt", 5120, 1024, 4096),
                  ("nn", 5120, 1024, 1024), ("nn", 5120, 1024, 3072), ("nn", 5120, 4096, 1024), ("nn", 5120, 1024, 4096),
                  ("tn", 3072, 1024, 5120), ("tn", 1024, 1024, 5120), ("tn", 4096, 1024, 5120), ("tn", 1024, 4096, 5120),
                  ("nt", 1024, 2048, 1024), ("nt", 1024, 1024, 2048), ("nn", 1024, 1024, 2048), ("nn", 1024, 2048, 1024),
                  ("tn", 2048, 1024, 1024), ("tn", 1024, 2048, 1024), ("nt", 1024, 1024, 384), ("nt", 1088, 3840, 1024),
                  ("nn", 1024, 2048, 6144), ("nn", 1024, 8192, 2048), ("nn", 1024, 2048, 8192), ("nn", 1024, 2048, 2048)]
    if os.environ.get("SHAPESET") == "pp_split":    # candidates for 256x256 tiles cut into K-slices: <= 85 tiles, long K
        shapes = [("nt", 5120, 1024, 4096), ("nn", 5120, 1024, 4096), ("nn", 5120, 1024, 3072), ("nt", 5120, 1024, 1024),
                  ("tn", 3072, 1024, 5120), ("tn", 4096, 1024, 5120), ("tn", 1024, 4096, 5120), ("tn", 1024, 1024, 5120),
                  ("tn", 2048, 2048, 5120), ("nn", 1024, 2048, 8192), ("nn", 1024, 2048, 6144), ("nt", 1024, 2048, 8192),
                  ("nt", 5120, 2048, 8192), ("nn", 5120, 2048, 8192), ("nn", 5120, 2048, 6144)]
        ops.set_workspace_bytes(256 << 20)
    variants = [int(v) for v in os.environ.get("VARIANTS", "1,3").split(",")]
    blas = os.environ.get("BLAS", "0") == "1"
    print(f"{'layout':6} {'M':>6} {'N':>6} {'K':>6} | " + " | ".join(f"v{v} ms    v{v} TF" for v in variants)
          + (" | blas ms  blas TF" if blas else ""))
    for lay, M, N, K in shapes:
        r = [bench(lay, M, N, K, v) for v in variants]
        if blas:
            r.append(bench_blas(lay, M, N, K))
        print(f"{lay:6} {M:6d} {N:6d} {K:6d} | " + " | ".join(f"{x[0]:8.4f} {x[1]:7.1f}" for x in r))

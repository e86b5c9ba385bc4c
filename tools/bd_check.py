"""B-direct GEMM kernels (csrc/gemm_bd.hip, variants 7 = 256x256 tiles, 8 = 160x256 tiles): correctness against fp64 torch on
NT shapes with tails / short K / epilogues, then timing against the ping-pong kernel (variant 3) and torch.mm (the vendor
library: a yard-stick only).  Usage: python tools/bd_check.py [check|bench|all]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afft_amd import _lib, ops  # noqa: E402

DEV = "cuda:0"


def setv(v):
    _lib.check(_lib.lib().afft_set_gemm_variant(v))


def pack_b(b):
    """row-major [N, K] bf16 -> fragment-major image (csrc/gemm_bd.hip, PACKED): [N/16][K/32][lane = (n & 15) + 16 ((k >> 3) & 3)][8];
    N is padded to a multiple of 16 with zero rows; returned as an [N16, K] tensor (same bytes)"""
    N, K = b.shape
    Np = (N + 15) // 16 * 16
    if Np != N:
        b = torch.cat([b, torch.zeros(Np - N, K, dtype=b.dtype, device=b.device)])
    return b.view(Np // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(Np, K)


def check():
    bad = 0
    g = torch.Generator().manual_seed(1)
    cases = [(160, 256, 64), (160, 256, 128), (160, 256, 192), (320, 512, 256), (5120, 2048, 512), (5120, 6144, 256),
             (200, 272, 320), (1000, 48, 448), (17, 16, 64), (4100, 2064, 1024), (256, 256, 64), (512, 768, 2048),
             (5120, 2048, 2048)]
    for variant in (8, 7, 10, 9):
        setv(variant)
        for (M, N, K) in cases:
            if variant >= 9 and (N % 16 or K % 64):
                continue
            a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(DEV)
            b = torch.randn(N, K, generator=g).to(torch.bfloat16).to(DEV)
            ref = a.double() @ b.double().t()
            if variant >= 9:
                b = pack_b(b)
            for mode in ("plain", "bias_res_f32", "bf16out"):
                if mode == "plain":
                    out = torch.empty(M, N, dtype=torch.float32, device=DEV)
                    ops.gemm(a, b, out, b_t=True)
                    r = ref
                elif mode == "bias_res_f32":
                    bias = torch.randn(N, generator=g).to(DEV)
                    res = torch.randn(M, N, generator=g).to(DEV)
                    out = torch.empty(M, N, dtype=torch.float32, device=DEV)
                    ops.gemm(a, b, out, b_t=True, bias=bias, residual=res, alpha=0.5)
                    r = 0.5 * ref + bias.double() + res.double()
                else:
                    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
                    ops.gemm(a, b, out, b_t=True)
                    r = ref
                torch.cuda.synchronize()
                err = ((out.double() - r).norm() / r.norm()).item()
                tol = 5e-3 if mode == "bf16out" else 2e-5
                ok = err < tol and bool(torch.isfinite(out).all())
                if not ok:
                    bad += 1
                    d = (out.double() - r).abs()
                    idx = (d > 1e-2 * r.abs().max()).nonzero()
                    print(f"  FAIL v{variant} {M}x{N}x{K} {mode}: rel {err:.3e}; bad elems {idx.shape[0]}; first {idx[:4].tolist()}")
            print(f"v{variant} {M}x{N}x{K} done", flush=True)
        # repeatability: the same launch 20 times gives the same bits (race screen)
        M, N, K = 5120, 2048, 2048
        a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(DEV)
        b = torch.randn(N, K, generator=g).to(torch.bfloat16).to(DEV)
        if variant >= 9:
            b = pack_b(b)
        first = None
        for it in range(20):
            out = torch.empty(M, N, dtype=torch.float32, device=DEV)
            ops.gemm(a, b, out, b_t=True)
            if first is None:
                first = out.clone()
            elif not torch.equal(first, out):
                bad += 1
                print(f"  FAIL v{variant}: launch {it} differs from launch 0 in {(first != out).sum().item()} elements")
                break
    setv(0)
    print("CHECK", "FAILED" if bad else "OK", bad)
    return bad


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def bench():
    shapes = [(5120, 2048, 2048), (5120, 6144, 2048), (5120, 8192, 2048), (5120, 2048, 8192), (5120, 2048, 6144),
              (8192, 8192, 8192), (8320, 8192, 8192), (1024, 2048, 2048), (1024, 8192, 2048)]
    variants = [3, 7, 8, 9, 10]
    g = torch.Generator().manual_seed(0)
    print(f"{'M':>6} {'N':>6} {'K':>6} | " + " | ".join(f"v{v} us   TF" for v in variants) + " | blas us  TF", flush=True)
    for rep in range(2):
        for (M, N, K) in shapes:
            a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(DEV)
            b = torch.randn(N, K, generator=g).to(torch.bfloat16).to(DEV)
            out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
            row = []
            bp = pack_b(b)
            for v in variants:
                setv(v)
                bb = bp if v >= 9 else b
                ms = timeit(lambda: ops.gemm(a, bb, out, b_t=True))
                row.append(ms)
            setv(0)
            bt = b.t()
            ms = timeit(lambda: torch.mm(a, bt))
            row.append(ms)
            fl = 2.0 * M * N * K
            print(f"{M:6d} {N:6d} {K:6d} | " + " | ".join(f"{x * 1e3:7.1f} {fl / x / 1e9:6.0f}" for x in row), flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    rc = 0
    if what in ("check", "all"):
        rc = check()
    if what in ("bench", "all"):
        bench()
    sys.exit(1 if rc else 0)

"""Times the HBM-bound kernels of the step in isolation at the cfg2 shapes and prints achieved GB/s (algorithmic
bytes / time).  Usage: python tools/ew_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afft_amd import ops  # noqa: E402

dev = "cuda:0"


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3   # us


def report(name, us, nbytes):
    print(f"{name:44s} {us:8.1f} us  {nbytes / us / 1e3:8.0f} GB/s")


def main():
    for rows in (5120, 1024):
        d = 2048
        x = torch.randn(rows, d, device=dev)
        w = torch.randn(d, device=dev)
        b = torch.randn(d, device=dev)
        y16 = torch.empty(rows, d, dtype=torch.bfloat16, device=dev)
        mean = torch.empty(rows, device=dev)
        rstd = torch.empty(rows, device=dev)
        report(f"ln_fwd {rows}x{d} f32->bf16", timeit(lambda: ops.layernorm_fwd(x, w, b, 1e-6, y16, mean, rstd)), rows * d * 6)
        dy16 = torch.randn(rows, d, device=dev).to(torch.bfloat16)
        dxin = torch.randn(rows, d, device=dev)
        dx = torch.empty(rows, d, device=dev)
        dw = torch.zeros(d, device=dev)
        db = torch.zeros(d, device=dev)
        report(f"ln_bwd {rows}x{d} (dy bf16, x, dx_in -> dx)",
               timeit(lambda: ops.layernorm_bwd(dy16, x, w, mean, rstd, dx, dx_in=dxin, dw=dw, db=db)), rows * d * 14)
        dx16 = torch.empty(rows, d, dtype=torch.bfloat16, device=dev)
        report(f"ln_bwd {rows}x{d} + bf16 copy of dx",
               timeit(lambda: ops.layernorm_bwd(dy16, x, w, mean, rstd, dx, dx_in=dxin, dx_bf16=dx16, dw=dw, db=db)), rows * d * 16)
        report(f"cast {rows}x{d} f32->bf16", timeit(lambda: ops.cast(x, y16)), rows * d * 6)
        g = torch.zeros(d, device=dev)
        report(f"colsum {rows}x{d} f32", timeit(lambda: ops.colsum(x, g)), rows * d * 4)
        for n in (2048, 6144, 8192):
            s16 = torch.randn(rows, n, device=dev).to(torch.bfloat16)
            g2 = torch.zeros(n, device=dev)
            report(f"colsum {rows}x{n} bf16", timeit(lambda: ops.colsum(s16, g2)), rows * n * 2)
        out = torch.zeros(16, d, device=dev)
        report(f"reduce_rows_periodic {rows}x{d} period 16", timeit(lambda: ops.reduce_rows_periodic(x, 16, out)), rows * d * 4)
        out5 = torch.zeros(5, d, device=dev)
        if rows % 5 == 0:
            report(f"reduce_rows_periodic {rows}x{d} period 5", timeit(lambda: ops.reduce_rows_periodic(x, 5, out5)), rows * d * 4)
    n = 32 * 1024 * 1024
    p = torch.randn(n, device=dev)
    gbuf = torch.randn(n, device=dev)
    buf = torch.zeros(n, device=dev)
    p16 = torch.empty(n, dtype=torch.bfloat16, device=dev)
    report("sgd 32M params (+bf16 image)", timeit(lambda: ops.sgd_nesterov(p, gbuf, buf, 0.01, 0.9, 1e-4, 1.0, False, p_bf16=p16)), n * 22)
    wmat = torch.randn(2048, 8192, device=dev)
    wt = torch.empty(8192, 2048, dtype=torch.bfloat16, device=dev)
    report("transpose-cast 2048x8192 f32->bf16", timeit(lambda: ops.cast(wmat, None, wt)), 2048 * 8192 * 6)


if __name__ == "__main__":
    main()

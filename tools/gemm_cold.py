"""Does a GEMM of the path run slower when its WEIGHT operand is cold (HBM) instead of resident in the Infinity Cache?
In a training step every weight is read once per pass and 1.2 GB of them pass through a 256 MB cache, so every GEMM sees cold
weights; tools/gemm_bench.py re-uses one buffer, so it sees warm ones.  For each shape: (a) one weight buffer re-used,
(b) 24 different weight buffers in rotation (> 256 MB in total), (c) as (b) with a read of the NEXT weight buffer issued on a
second stream while the current GEMM runs (software prefetch into the Infinity Cache).
usage: python tools/gemm_cold.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afft_amd import ops  # noqa: E402

dev = "cuda:0"
NW = 24


def run(layout, M, N, K, mode, iters=48):
    g = torch.Generator().manual_seed(0)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    wshape = (N, K) if layout == "nt" else (K, N)
    ws = [torch.randn(wshape, generator=g).to(torch.bfloat16).to(dev) for _ in range(NW if mode != "warm" else 1)]
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    kw = dict(b_t=True) if layout == "nt" else {}
    side = torch.cuda.Stream()
    sink = torch.zeros(1, device=dev)
    for i in range(6):
        ops.gemm(a, ws[i % len(ws)], out, **kw)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters):
        if mode == "prefetch":
            side.wait_stream(torch.cuda.current_stream())          # not before the previous GEMM has been issued
            with torch.cuda.stream(side):
                sink += ws[(i + 1) % NW].view(torch.int32)[::1, :].sum()     # reads the whole buffer once
        ops.gemm(a, ws[i % len(ws)], out, **kw)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    return ms, 2.0 * M * N * K / ms / 1e9


if __name__ == "__main__":
    shapes = [("nn", 1024, 6144, 2048), ("nn", 1024, 8192, 2048), ("nn", 1024, 2048, 8192), ("nn", 1024, 2048, 2048),
              ("nt", 1024, 8192, 2048), ("nt", 1024, 2048, 8192),
              ("nt", 5120, 6144, 2048), ("nt", 5120, 2048, 2048), ("nt", 5120, 8192, 2048), ("nt", 5120, 2048, 8192),
              ("nn", 5120, 2048, 6144), ("nn", 5120, 8192, 2048), ("nn", 5120, 2048, 8192)]
    print(f"{'layout':6} {'M':>6} {'N':>6} {'K':>6} | warm ms    TF | cold ms    TF | prefetch ms  TF")
    for lay, M, N, K in shapes:
        r = [run(lay, M, N, K, m) for m in ("warm", "cold", "prefetch")]
        print(f"{lay:6} {M:6d} {N:6d} {K:6d} | " + " | ".join(f"{x[0]:8.4f} {x[1]:7.1f}" for x in r))

"""Does the optimizer epilogue's speed depend on WHERE its buffers lie?  (round 6: per process, the fused weight-gradient + SGD launches of the
step take ~52 us or ~104 us.)  One process, the K = 1024 weight gradient of a 8192 x 2048 weight with the update in its epilogue, over `n`
independently allocated (parameter, momentum, image) triples: per triple the median launch time and the buffers' device addresses.
usage: python tools/addr_probe.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from afft_amd import _lib, ops

dev = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
M, N, K = 8192, 2048, 1024
a = torch.randn(K, M, device=dev).to(torch.bfloat16)
b = torch.randn(K, N, device=dev).to(torch.bfloat16)
g = torch.empty(M, N, device=dev)
keep = []
rows = []
for i in range(n):
    if i % 3 == 1:
        keep.append(torch.empty((i * 37 + 5) << 20, dtype=torch.uint8, device=dev))      # perturb the allocator's layout
    p = torch.randn(M, N, device=dev)
    m = torch.zeros(M, N, device=dev)
    p16 = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    d = _lib.SgdFused()
    d.p, d.buf, d.p_bf16, d.lr, d.mom, d.wd, d.gscale, d.first_step = p.data_ptr(), m.data_ptr(), p16.data_ptr(), 1e-3, 0.9, 1e-6, 1.0, 0
    ts = []
    for r in range(7):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(4):
            ops.gemm(a, b, g, a_t=True, sgd=d)
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / 4 * 1e3)
    ts.sort()
    rows.append((ts[3], p.data_ptr(), m.data_ptr(), p16.data_ptr()))
    keep.append((p, m, p16))
for t, pp, mm, ii in rows:
    print(f"{t:7.1f} us  p {pp:#x} mom {mm:#x} img {ii:#x}  mom-p {(mm - pp) >> 20} MiB  p mod 1GiB {(pp & ((1 << 30) - 1)) >> 20} MiB")

"""Does the row pitch of the operands matter (L2 / memory channel interleave)?  NT GEMMs with padded leading dimensions.
Usage: python tools/gemm_pitch.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afft_amd import _lib, ops  # noqa: E402
dev = "cuda:0"


def run(lay, M, N, K, pad_a, pad_b, variant=3, iters=20):
    _lib.check(_lib.lib().afft_set_gemm_variant(variant))
    def mk(r, c, pad):
        buf = torch.randn(r, c + pad, device=dev).to(torch.bfloat16)
        return buf[:, :c]
    if lay == "nt":
        a, b, kw = mk(M, K, pad_a), mk(N, K, pad_b), dict(b_t=True)
    elif lay == "nn":
        a, b, kw = mk(M, K, pad_a), mk(K, N, pad_b), dict()
    else:
        a, b, kw = mk(K, M, pad_a), mk(K, N, pad_b), dict(a_t=True)
    out = torch.empty(M, N, dtype=torch.bfloat16 if lay != "tn" else torch.float32, device=dev)
    for _ in range(3):
        ops.gemm(a, b, out, **kw)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        ops.gemm(a, b, out, **kw)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


pads = [(0, 0), (64, 64), (128, 128), (32, 32), (64, 0), (0, 64), (192, 192)]
print("layout M N K | us at (pad_a, pad_b) in elements: " + " ".join(str(p) for p in pads))
for lay, M, N, K in [("nt", 5120, 6144, 2048), ("nt", 5120, 2048, 8192), ("nt", 5120, 8192, 2048), ("nn", 5120, 2048, 6144),
                     ("nn", 5120, 8192, 2048), ("tn", 8192, 2048, 5120), ("tn", 6144, 2048, 5120), ("nt", 8192, 8192, 8192),
                     ("nt", 1024, 6144, 2048)]:
    v = 3 if M > 1024 else 1
    print(lay, M, N, K, "|", "  ".join("%7.1f" % run(lay, M, N, K, pa, pb, v) for pa, pb in pads))
_lib.check(_lib.lib().afft_set_gemm_variant(0))

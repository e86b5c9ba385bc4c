"""What do the fused epilogues cost?  The path's GEMM shapes with the epilogue they carry in the step (bias + GELU + two outputs,
data gradient x GELU' with the pre-activation as aux, bias + dropout + residual into fp32) against the plain product, per tile
shape.  usage: python tools/gemm_epi.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afft_amd import _lib, ops  # noqa: E402

dev = "cuda:0"


def run(layout, M, N, K, epi, variant, iters=30):
    _lib.check(_lib.lib().afft_set_gemm_variant(variant))
    g = torch.Generator().manual_seed(0)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    b = (torch.randn(N, K, generator=g) if layout == "nt" else torch.randn(K, N, generator=g)).to(torch.bfloat16).to(dev)
    kw = dict(b_t=True) if layout == "nt" else {}
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    if epi == "gelu_erf+pre":
        kw.update(bias=torch.randn(N, device=dev), act=ops.ACT_GELU_ERF, pre=torch.empty(M, N, dtype=torch.bfloat16, device=dev))
    elif epi == "gelu_erf":
        kw.update(bias=torch.randn(N, device=dev), act=ops.ACT_GELU_ERF)
    elif epi == "bias+pre":
        kw.update(bias=torch.randn(N, device=dev), pre=torch.empty(M, N, dtype=torch.bfloat16, device=dev))
    elif epi == "bias":
        kw.update(bias=torch.randn(N, device=dev))
    elif epi == "gelu_tanh+pre":
        kw.update(bias=torch.randn(N, device=dev), act=ops.ACT_GELU_TANH, pre=torch.empty(M, N, dtype=torch.bfloat16, device=dev))
    elif epi == "dgelu_erf":
        kw.update(act=ops.ACT_DGELU_ERF, aux=torch.randn(M, N, device=dev).to(torch.bfloat16))
    elif epi == "dgelu_tanh":
        kw.update(act=ops.ACT_DGELU_TANH, aux=torch.randn(M, N, device=dev).to(torch.bfloat16))
    elif epi == "bias+res_f32":
        out = torch.empty(M, N, dtype=torch.float32, device=dev)
        kw.update(bias=torch.randn(N, device=dev), residual=torch.randn(M, N, device=dev))
    for _ in range(3):
        ops.gemm(a, b, out, **kw)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        ops.gemm(a, b, out, **kw)
    e.record()
    torch.cuda.synchronize()
    _lib.check(_lib.lib().afft_set_gemm_variant(0))
    return s.elapsed_time(e) / iters * 1e3


if __name__ == "__main__":
    cases = [("nt", 5120, 8192, 2048, "gelu_erf+pre"), ("nn", 5120, 8192, 2048, "dgelu_erf"), ("nt", 5120, 2048, 8192, "bias+res_f32"),
             ("nt", 5120, 2048, 2048, "bias+res_f32"), ("nn", 1024, 8192, 2048, "gelu_tanh+pre"), ("nt", 1024, 8192, 2048, "dgelu_tanh"),
             ("nn", 1024, 2048, 8192, "bias+res_f32"), ("nt", 5120, 4096, 1024, "gelu_erf+pre"), ("nn", 5120, 4096, 1024, "dgelu_erf")]
    if os.environ.get("DECOMPOSE") == "1":
        cases = [("nt", 5120, 8192, 2048, e) for e in ("bias", "gelu_erf", "bias+pre", "gelu_erf+pre")] + [("nn", 5120, 8192, 2048, "dgelu_erf")]
    print(f"{'layout':6} {'M':>6} {'N':>6} {'K':>6} {'epilogue':>14} | 128x128: plain us  with epilogue | 256x256: plain us  with epilogue")
    for lay, M, N, K, epi in cases:
        r = [run(lay, M, N, K, e, v) for v in (1, 3) for e in ("plain", epi)]
        print(f"{lay:6} {M:6d} {N:6d} {K:6d} {epi:>14} | {r[0]:17.1f} {r[1]:14.1f} | {r[2]:17.1f} {r[3]:14.1f}")

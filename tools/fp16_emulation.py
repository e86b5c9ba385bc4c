"""Would fp16 operands (11-bit significands, fp32 accumulation: v_mfma_f32_16x16x32_f16) put a single-pass forward inside the
north-star tolerance of 1e-3?    Emulation on the exact-fp32 path: every GEMM operand and every attention
input is rounded to the 16-bit format and back before the exact-fp32 kernels run -- what a 16-bit-operand MFMA with fp32
accumulation computes, up to summation order.  The bf16 rows check the emulation against the real bf16 mode of the library.
Optimistic on one point: the attention probabilities stay fp32 (the MFMA kernels round them to 16 bits for P.V).
usage: python tools/fp16_emulation.py [config] [clips]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
import afft_amd  # noqa: E402
from afft_amd import ops  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
nclips = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
afft_amd.set_grad_mode("sink")


def logits(mode, rounder=None, which="ab"):
    afft_amd.set_precision(mode)
    model, c = B.build_model(name, dev)
    model.eval()
    feats, tgt, sub = B.make_inputs(c, nclips, c["T"], 0, dev)
    real_gemm, real_attn = ops.gemm, ops.attention_fwd
    stats = {"max_abs": 0.0}
    if rounder is not None:
        def r(t):
            if isinstance(t, torch.Tensor) and t.dtype == torch.float32:
                stats["max_abs"] = max(stats["max_abs"], float(t.abs().max()))
                return rounder(t)
            return t

        def gemm(a, b, out, **kw):      # forward GEMMs: a = activation, b = weight
            return real_gemm(r(a) if "a" in which else a, r(b) if "b" in which else b, out, **kw)

        def attn(q, k, v, *a, **kw):
            if "a" not in which:
                return real_attn(q, k, v, *a, **kw)
            return real_attn(r(q), r(k), r(v), *a, **kw)
        ops.gemm, ops.attention_fwd = gemm, attn
    try:
        with torch.no_grad():
            o, _ = model(feats, mixup_fn=None, target=tgt, target_subclips=sub, target_subclips_ignore_index=None)
    finally:
        ops.gemm, ops.attention_fwd = real_gemm, real_attn
    torch.cuda.synchronize()
    return {k: o[k]["all-fused"].double() for k in ("logits/action", "past_logits/action", "future")}, stats["max_abs"]


def err(a, b):
    return {k: float(((a[k] - b[k]).norm() / b[k].norm()).cpu()) for k in b}


ref, _ = logits("fp32")
rows = [("bf16 mode of the library (real kernels)", logits("bf16")[0], None),
        ("bf16x3 mode of the library (real kernels)", logits("bf16x3")[0], None)]
e_bf, m_bf = logits("fp32", lambda t: t.to(torch.bfloat16).to(torch.float32))
rows.append(("EMULATED bf16 operands on the exact path", e_bf, m_bf))
e_h, m_h = logits("fp32", lambda t: t.to(torch.float16).to(torch.float32))
rows.append(("EMULATED fp16 operands on the exact path", e_h, m_h))
h16 = lambda t: t.to(torch.float16).to(torch.float32)      # noqa: E731
rows.append(("EMULATED fp16 weights, exact activations", *logits("fp32", h16, "b")))
rows.append(("EMULATED fp16 activations, exact weights", *logits("fp32", h16, "a")))
print(f"{name}, {nclips} clips, eval mode: relative L2 error against the exact-fp32 mode (north star: 1e-3 on the logits)")
print(f"{'operands':46} {'logits/action':>14} {'past_logits':>12} {'future':>10} {'max |operand|':>14}")
for label, o, mx in rows:
    e = err(o, ref)
    print(f"{label:46} {e['logits/action']:14.2e} {e['past_logits/action']:12.2e} {e['future']:10.2e} {('%.1f' % mx) if mx else '':>14}")
print("fp16 overflows at 65504; the largest operand magnitude above shows the head-room on this (random-init, synthetic) workload.")

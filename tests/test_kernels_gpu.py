"""GPU: every C-ABI kernel against plain fp32/fp64 torch math on the same seeded inputs.

Tolerances: fp32 kernels 2e-5 relative-L2 (exact-fp32 MFMA, different summation order only);
bf16-operand kernels 1e-2 (bf16 rounding of the inputs is applied to the reference too, so what is
left is accumulation order + output rounding)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import rel_l2  # noqa: E402


def dev():
    return torch.device("cuda:0")


def rnd(*shape, seed=0, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def bfr(t):  # round to bf16 and back (reference sees the same operand values)
    return t.to(torch.bfloat16).to(torch.float32)


def _act(kind, v, aux):
    from oracle import afft_oracle as O
    if kind == 0:
        return v
    if kind == 1:
        return O.gelu_erf(v)
    if kind == 2:
        return O.gelu_tanh(v)
    x = aux.clone().requires_grad_(True)
    (O.gelu_erf(x) if kind == 3 else O.gelu_tanh(x)).sum().backward()
    return v * x.grad


GEMM_CASES = [
    # name, M, N, K, dtype, layout, epilogue
    ("nt_bf16_full", 256, 384, 256, "bf16", "nt", dict()),
    ("nt_bf16_tails", 200, 3806, 128, "bf16", "nt", dict(bias=True, out_f32=True, ldo_pad=3840)),
    ("nt_bf16_gelu", 130, 512, 192, "bf16", "nt", dict(bias=True, act=1, pre=True)),
    ("nt_bf16_tanh_res", 128, 256, 64, "bf16", "nt", dict(bias=True, act=2)),
    ("nt_bf16_residual", 320, 256, 512, "bf16", "nt", dict(bias=True, residual=True, out_f32=True, rowscale=True)),
    ("nt_bf16_dgelu", 128, 256, 128, "bf16", "nt", dict(act=3)),
    ("nt_bf16_dgelu_tanh", 64, 128, 64, "bf16", "nt", dict(act=4, out2=True)),
    ("tn_bf16_full", 256, 384, 256, "bf16", "tn", dict(out_f32=True)),
    ("tn_bf16_tails_acc", 3806, 200, 192, "bf16", "tn", dict(out_f32=True, accumulate=True, lda_pad=3840, ldb_pad=256)),
    ("tn_bf16_small", 24, 64, 64, "bf16", "tn", dict(out_f32=True)),
    ("nt_bf16_fallback_k", 70, 50, 24, "bf16", "nt", dict(bias=True, out_f32=True)),   # K % 64 != 0 -> fp32-MFMA path
    # small grids: split-K (K >= 2048: 2 slices, K >= 6144: 4), the last-arriving slice sums and runs the epilogue
    ("nt_bf16_splitk2", 256, 384, 2048, "bf16", "nt", dict(bias=True, residual=True, out_f32=True, rowscale=True)),
    ("nn_bf16_splitk4", 200, 256, 8192, "bf16", "nn", dict(bias=True, out_f32=True)),
    ("nt_bf16_splitk2_gelu", 256, 256, 2048, "bf16", "nt", dict(bias=True, act=1, pre=True)),
    ("tn_bf16_splitk4_acc", 200, 130, 6144, "bf16", "tn", dict(out_f32=True, accumulate=True, lda_pad=256, ldb_pad=192)),
    ("nt_bf16_nosplit_gelu", 256, 256, 2048, "bf16", "nt", dict(bias=True, act=1)),
    ("nt_bf16_pp", 300, 520, 192, "bf16", "nt", dict(bias=True, act=2, out2=True)),
    ("nt_f32", 150, 130, 100, "f32", "nt", dict(bias=True, act=1, pre=True)),
    ("nn_f32", 96, 200, 77, "f32", "nn", dict(bias=True, residual=True, out_f32=True)),
    ("tn_f32", 66, 70, 130, "f32", "tn", dict(accumulate=True, out_f32=True)),
    ("tt_f32", 33, 65, 17, "f32", "tt", dict(out_f32=True, alpha=0.5)),
    ("nt_f32_dgelu", 64, 64, 64, "f32", "nt", dict(act=4)),
]



@pytest.mark.parametrize("case", GEMM_CASES, ids=[c[0] for c in GEMM_CASES])
def test_gemm(case):
    from afft_amd import ops
    name, M, N, K, dt, layout, ep = case
    from afft_amd import _lib
    _lib.check(_lib.lib().afft_set_gemm_splitk(0 if "nosplit" in name else 1))   # 1 = auto (default)
    _lib.check(_lib.lib().afft_set_gemm_variant(3 if "_pp" in name else 0))
    tdt = torch.bfloat16 if dt == "bf16" else torch.float32
    a_t, b_t = layout[0] == "t", layout[1] == "t"
    lda_pad, ldb_pad = ep.get("lda_pad"), ep.get("ldb_pad")
    A = rnd(M, K, seed=1)            # logical A
    Bm = rnd(K, N, seed=2)           # logical B
    if dt == "bf16":
        A, Bm = bfr(A), bfr(Bm)

    def store(logical, transposed, pad):
        t = logical.t().contiguous() if transposed else logical.contiguous()
        if pad:
            buf = torch.zeros(t.shape[0], pad)
            buf[:, :t.shape[1]] = t
            return buf.to(tdt).to(dev())[:, :t.shape[1]]
        return t.to(tdt).to(dev())

    a_store = store(A, a_t, lda_pad)
    # B storage: 'n' second letter => stored [K,N]; 't' => stored [N,K]
    b_store = store(Bm, b_t, ldb_pad)
    bias = rnd(N, seed=3) if ep.get("bias") else None
    act = ep.get("act", 0)
    aux = rnd(M, N, seed=4) if act >= 3 else None
    if aux is not None and dt == "bf16":
        aux = bfr(aux)
    res = rnd(M, N, seed=5) if ep.get("residual") else None
    rowscale = (rnd(M, seed=6).abs() + 0.5) if ep.get("rowscale") else None
    out_dt = torch.float32 if (ep.get("out_f32") or dt == "f32") else torch.bfloat16
    ldo = ep.get("ldo_pad", N)
    out_buf = torch.zeros(M, ldo, dtype=out_dt, device=dev())
    out = out_buf[:, :N]
    init = None
    if ep.get("accumulate"):
        init = rnd(M, N, seed=7)
        out.copy_(init.to(dev()))
    pre = torch.zeros(M, N, dtype=tdt, device=dev()) if ep.get("pre") else None
    out2 = torch.zeros(M, N, dtype=torch.float32 if out_dt == torch.bfloat16 else torch.bfloat16, device=dev()) if ep.get("out2") else None
    alpha = ep.get("alpha", 1.0)
    ops.gemm(a_store, b_store, out, a_t=a_t, b_t=b_t, bias=None if bias is None else bias.to(dev()), act=act,
             aux=None if aux is None else aux.to(tdt).to(dev()), pre=pre,
             rowscale=None if rowscale is None else rowscale.to(dev()),
             residual=None if res is None else res.to(dev()), accumulate=bool(ep.get("accumulate")), out2=out2,
             alpha=alpha)
    torch.cuda.synchronize()
    ref = alpha * (A.double() @ Bm.double()).float()
    if bias is not None:
        ref = ref + bias
    pre_ref = ref.clone()
    ref = _act(act, ref, aux)
    if rowscale is not None:
        ref = ref * rowscale[:, None]
    if res is not None:
        ref = ref + res
    if init is not None:
        ref = ref + init
    tol = 2e-5 if (dt == "f32") else (1e-2 if out_dt == torch.bfloat16 else 2e-3)
    if name == "nt_bf16_fallback_k":
        tol = 2e-5
    assert rel_l2(out.float().cpu(), ref) < tol, name
    if pre is not None:
        assert rel_l2(pre.float().cpu(), pre_ref) < (2e-5 if dt == "f32" else 1e-2)
    if out2 is not None:
        assert rel_l2(out2.float().cpu(), ref) < 1e-2
    if ldo != N:  # padding columns untouched
        assert float(out_buf[:, N:].abs().max()) == 0.0
    if "splitk" in name and not ep.get("accumulate"):   # the sum is taken in slice order whoever arrives last: bitwise repeatable
        first = out.clone()
        for _ in range(3):
            ops.gemm(a_store, b_store, out, a_t=a_t, b_t=b_t, bias=None if bias is None else bias.to(dev()), act=act,
                     aux=None if aux is None else aux.to(tdt).to(dev()), pre=pre,
                     rowscale=None if rowscale is None else rowscale.to(dev()),
                     residual=None if res is None else res.to(dev()), out2=out2, alpha=alpha)
            assert torch.equal(out, first)
    _lib.check(_lib.lib().afft_set_gemm_splitk(1))
    _lib.check(_lib.lib().afft_set_gemm_variant(0))


def test_split_bf16_planes():
    """afft_split_bf16: x = hi + lo to ~2^-17 relative, zero tails out to the padded plane."""
    from afft_amd import ops
    x = rnd(70, 100, seed=11) * 3.0
    sp = ops.Split(x.to(dev()))
    torch.cuda.synchronize()
    assert sp.planes.shape == (2, 128, 128)
    hi, lo = sp.planes[0].float().cpu(), sp.planes[1].float().cpu()
    assert torch.equal(hi[:70, :100], bfr(x))
    assert float((hi + lo)[:70, :100].sub(x).abs().max()) <= float(x.abs().max()) * 2.0 ** -16
    assert float(hi[70:].abs().max()) == 0 and float(hi[:, 100:].abs().max()) == 0
    assert float(lo[70:].abs().max()) == 0 and float(lo[:, 100:].abs().max()) == 0
    # a strided (non-16-byte-aligned rows) source takes the scalar path
    y = rnd(33, 67, seed=12).to(dev())
    sp2 = ops.Split(y[:, 1:66])
    torch.cuda.synchronize()
    assert torch.equal(sp2.planes[0].float().cpu()[:33, :65], bfr(y[:, 1:66].cpu()))


X3_CASES = [
    # name, M, N, K, layout, variant (0 = auto: 128x128 kernel at these sizes, 3 = 256x256 ping-pong forced), epilogue
    ("nt_x3", 300, 520, 200, "nt", 0, dict(bias=True, act=1, pre=True)),
    ("nn_x3", 130, 256, 320, "nn", 0, dict(bias=True, residual=True)),
    ("tn_x3", 256, 130, 500, "tn", 0, dict(accumulate=True)),
    ("nt_x3_pp", 300, 520, 200, "nt", 3, dict(bias=True, act=2)),
    ("nn_x3_pp", 512, 300, 320, "nn", 3, dict(residual=True)),
    ("tn_x3_pp", 512, 264, 704, "tn", 3, dict(accumulate=True)),
    ("nt_x3_big", 1100, 3806, 2048, "nt", 0, dict(bias=True)),      # auto-selected ping-pong kernel, classifier-shaped tails
    # whole 256x256 tiles with an even K-tile count per segment: the steady-state kernel (gemm_bf16_pp2_kernel<*, *, 1>), whose LDS-DMA stream
    # JUMPS from one pair of operand planes to the next where a segment ends -- a wrong jump reads a hi plane for a lo one: 1e-3, not 1e-5
    ("nt_x3_pp2", 512, 768, 256, "nt", 3, dict(bias=True, act=1, pre=True)),
    ("nn_x3_pp2", 512, 512, 384, "nn", 3, dict(residual=True)),
    ("tn_x3_pp2", 512, 256, 640, "tn", 3, dict(accumulate=True)),
    ("nt_x3_pp2_two_ktiles", 256, 256, 128, "nt", 3, dict()),       # one pair per segment: the jump sits in the first loop iteration
]


@pytest.mark.parametrize("case", X3_CASES, ids=[c[0] for c in X3_CASES])
def test_gemm_bf16x3(case):
    """bf16x3: A_hi B_hi + A_lo B_hi + A_hi B_lo in one launch over two-plane splits of fp32 operands -- fp32-grade results
    (the lo x lo term, 2^-18 relative, is the only thing dropped) from the bf16 MFMA kernels, every layout and both tiles."""
    from afft_amd import _lib, ops
    name, M, N, K, layout, variant, ep = case
    _lib.check(_lib.lib().afft_set_gemm_variant(variant))
    A, Bm = rnd(M, K, seed=21), rnd(K, N, seed=22)
    a_t, b_t = layout[0] == "t", layout[1] == "t"
    sa = ops.Split((A.t().contiguous() if a_t else A).to(dev()))
    sb = ops.Split((Bm.t().contiguous() if b_t else Bm).to(dev()))
    bias = rnd(N, seed=3) if ep.get("bias") else None
    act = ep.get("act", 0)
    res = rnd(M, N, seed=5) if ep.get("residual") else None
    out = torch.zeros(M, N, device=dev())
    init = None
    if ep.get("accumulate"):
        init = rnd(M, N, seed=7)
        out.copy_(init.to(dev()))
    pre = torch.zeros(M, N, device=dev()) if ep.get("pre") else None
    ops.gemm(sa, sb, out, a_t=a_t, b_t=b_t, bias=None if bias is None else bias.to(dev()), act=act, pre=pre,
             residual=None if res is None else res.to(dev()), accumulate=bool(ep.get("accumulate")))
    torch.cuda.synchronize()
    _lib.check(_lib.lib().afft_set_gemm_variant(0))
    ref = (A.double() @ Bm.double()).float()
    if bias is not None:
        ref = ref + bias
    pre_ref = ref.clone()
    ref = _act(act, ref, None)
    if res is not None:
        ref = ref + res
    if init is not None:
        ref = ref + init
    assert rel_l2(out.cpu(), ref) < 1e-5, (name, rel_l2(out.cpu(), ref))
    if pre is not None:
        assert rel_l2(pre.cpu(), pre_ref) < 1e-5


F16X2_CASES = [
    ("nt_f16x2", 300, 520, 200, "nt", 0, dict(bias=True, act=1, pre=True)),
    ("nn_f16x2", 130, 256, 320, "nn", 0, dict(bias=True, residual=True)),
    ("nt_f16x2_pp", 300, 520, 200, "nt", 3, dict(bias=True, act=2)),
    ("nn_f16x2_pp", 512, 300, 320, "nn", 3, dict(residual=True)),
    ("nt_f16x2_big", 1100, 3806, 2048, "nt", 0, dict(bias=True)),
    ("nt_f16x2_pp2", 512, 768, 256, "nt", 3, dict(bias=True, act=2)),      # whole tiles: gemm_bf16_pp2_kernel<false, *, 2>
    ("nn_f16x2_pp2", 256, 512, 384, "nn", 3, dict(residual=True)),
]


@pytest.mark.parametrize("case", F16X2_CASES, ids=[c[0] for c in F16X2_CASES])
def test_gemm_fp16x2(case):
    """fp16 two-pass forward GEMM (afft_gemm_t.split3 = 2): A = hi + lo in fp16 (exact to 2^-22), B rounded ONCE to fp16, two
    segments A_hi B + A_lo B on v_mfma_f32_16x16x32_f16.  Against float64 with B rounded to fp16 the result is fp32-grade; against
    the exact product the error is B's rounding, 2^-11 / sqrt(3) per element ~ 2e-4 relative -- 20x below the bf16 GEMM's."""
    from afft_amd import _lib, ops
    name, M, N, K, layout, variant, ep = case
    _lib.check(_lib.lib().afft_set_gemm_variant(variant))
    A, Bm = rnd(M, K, seed=31), rnd(K, N, seed=32)
    b_t = layout[1] == "t"
    sa = ops.Split(A.to(dev()), f16=True)
    sb = ops.Split((Bm.t().contiguous() if b_t else Bm).to(dev()), f16=True)
    assert sa.planes.dtype == torch.float16
    torch.cuda.synchronize()
    assert torch.equal(sa.planes[0][:M, :K].cpu(), A.half())
    assert float((sa.planes[0].float() + sa.planes[1].float())[:M, :K].cpu().sub(A).abs().max()) <= float(A.abs().max()) * 2.0 ** -21
    bias = rnd(N, seed=3) if ep.get("bias") else None
    act = ep.get("act", 0)
    res = rnd(M, N, seed=5) if ep.get("residual") else None
    out = torch.zeros(M, N, device=dev())
    pre = torch.zeros(M, N, device=dev()) if ep.get("pre") else None
    ops.gemm(sa, sb, out, b_t=b_t, bias=None if bias is None else bias.to(dev()), act=act, pre=pre,
             residual=None if res is None else res.to(dev()))
    torch.cuda.synchronize()
    _lib.check(_lib.lib().afft_set_gemm_variant(0))

    def finish(prod):
        r = prod.float()
        if bias is not None:
            r = r + bias
        r = _act(act, r, None)
        return r + res if res is not None else r
    rounded, exact = finish(A.double() @ Bm.half().double()), finish(A.double() @ Bm.double())
    assert rel_l2(out.cpu(), rounded) < 1e-5, (name, rel_l2(out.cpu(), rounded))
    e = rel_l2(out.cpu(), exact)
    assert 2e-5 < e < 4e-4, (name, e)
    # backward layouts are not built for this mode: a clear error, not a wrong kernel
    with pytest.raises(RuntimeError, match="forward layouts only|fast-path layout"):
        ops.gemm(ops.Split(A.t().contiguous().to(dev()), f16=True), sb, out, a_t=True, b_t=b_t)
    with pytest.raises(TypeError, match="fp16 planes"):
        ops.gemm(sa, ops.Split((Bm.t().contiguous() if b_t else Bm).to(dev())), out, b_t=b_t)
    # B as the weight's plain FP16 image (what the training forward passes): the same result, bit for bit
    out_b = torch.zeros(M, N, device=dev())
    ops.gemm(sa, sb.planes[0][:N] if b_t else sb.planes[0][:, :N], out_b, b_t=b_t, bias=None if bias is None else bias.to(dev()), act=act,
             residual=None if res is None else res.to(dev()))
    torch.cuda.synchronize()
    assert torch.equal(out_b, out)


ONE_PASS_CASES = [
    ("nt_pp2", 512, 768, 256, "nt", 3, dict(bias=True, act=2)),        # whole tiles, 4 K-tiles: gemm_bf16_pp2_kernel<false, false, 2>, one segment
    ("nt_pp2_fc2", 5120, 2048, 8192, "nt", 0, dict(bias=True, residual=True)),      # the fusers' fc2 at cfg2 (runtime.one_pass_sites default)
    ("nn_pp2", 256, 512, 384, "nn", 3, dict(residual=True)),
    ("nt_pp_tail", 300, 520, 192, "nt", 3, dict(bias=True)),            # ragged: the general 256x256 kernel
    ("nt_pp_2tiles", 512, 512, 128, "nt", 3, dict()),                   # 2 K-tiles: below the steady-state kernel's minimum
    ("nn_128_predictor", 1024, 2048, 2048, "nn", 0, dict(bias=True)),   # the predictor's Conv1D shapes (128x128 kernel, split-K by the cost model)
    ("nn_128_fc2", 1024, 2048, 8192, "nn", 0, dict(bias=True, residual=True)),
    ("nn_128_tail", 130, 256, 320, "nn", 1, dict(bias=True, act=1, pre=True)),
]


@pytest.mark.parametrize("case", ONE_PASS_CASES, ids=[c[0] for c in ONE_PASS_CASES])
def test_gemm_fp16_one_pass(case):
    """ONE fp16 pass (afft_gemm_t.split3 = 4, the AFFT_F16X2_ONE_PASS_* sites of the sub-layers): the activation's hi plane against the
    weight's FP16 image on the fp16 MFMA, one segment of the two-pass kernels.  Against float64 on the fp16-rounded operands the result
    is fp32-grade; against the exact product it carries both roundings (~2.4e-4 relative)."""
    from afft_amd import _lib, ops
    name, M, N, K, layout, variant, ep = case
    A, Bm = rnd(M, K, seed=41), rnd(K, N, seed=42)
    b_t = layout[1] == "t"
    a16 = A.half().to(dev())
    b16 = (Bm.t().contiguous() if b_t else Bm).half().to(dev())
    bias = rnd(N, seed=3) if ep.get("bias") else None
    act = ep.get("act", 0)
    res = rnd(M, N, seed=5) if ep.get("residual") else None
    out = torch.zeros(M, N, device=dev())
    pre = torch.zeros(M, N, device=dev()) if ep.get("pre") else None
    _lib.check(_lib.lib().afft_set_gemm_variant(variant))
    ops.gemm(a16, b16, out, b_t=b_t, bias=None if bias is None else bias.to(dev()), act=act, pre=pre,
             residual=None if res is None else res.to(dev()))
    torch.cuda.synchronize()
    _lib.check(_lib.lib().afft_set_gemm_variant(0))

    def finish(prod):
        r = prod.float()
        if bias is not None:
            r = r + bias
        r = _act(act, r, None)
        return r + res if res is not None else r
    rounded, exact = finish(A.half().double() @ Bm.half().double()), finish(A.double() @ Bm.double())
    assert rel_l2(out.cpu(), rounded) < 1e-5, (name, rel_l2(out.cpu(), rounded))
    if not ep.get("residual"):
        assert 5e-5 < rel_l2(out.cpu(), exact) < 6e-4, (name, rel_l2(out.cpu(), exact))
    with pytest.raises(RuntimeError, match="forward layouts only|fast-path layout"):
        ops.gemm(a16.t().contiguous(), b16, torch.zeros(M, N, device=dev()), a_t=True, b_t=b_t)


@pytest.mark.parametrize("case", [("nt_pp", 512, 768, 256, "nt", 3), ("nn_128", 130, 256, 320, "nn", 1), ("nt_tail", 300, 520, 200, "nt", 0)],
                         ids=lambda c: c[0])
def test_gemm_fp16x2_plane_output_feeds_the_next_gemm(case):
    """The fp16x2 training forward: a GEMM whose B is a weight's plain FP16 image writes its result as the operand planes of the
    next GEMM (afft_gemm_t.out_lo: hi = fp16(v), lo = fp16(v - hi)), the bf16 copy the backward pass reads (out2) and the bf16
    pre-activation (pre).  hi / lo are bitwise what splitting the fp32 result gives; hi + lo is the result to 2^-21."""
    from afft_amd import _lib, ops
    name, M, N, K, layout, variant = case
    _lib.check(_lib.lib().afft_set_gemm_variant(variant))
    A, Bm = rnd(M, K, seed=41), rnd(K, N, seed=42)
    b_t = layout[1] == "t"
    sa = ops.Split(A.to(dev()), f16=True)
    pN, pM, pK = (N + 63) // 64 * 64, (M + 63) // 64 * 64, (K + 63) // 64 * 64
    if b_t:
        Bh = torch.zeros(N, pK, dtype=torch.float16, device=dev())
        Bh[:, :K] = Bm.t().half().to(dev())
    else:
        Bh = torch.zeros(pK, N, dtype=torch.float16, device=dev())
        Bh[:K] = Bm.half().to(dev())
    bias = rnd(N, seed=3).to(dev())
    ref = torch.zeros(M, N, device=dev())
    ops.gemm(sa, Bh, ref, b_t=b_t, bias=bias, act=1)
    planes = torch.zeros(2, pM, pN, dtype=torch.float16, device=dev())
    copy = torch.zeros(pM, pN, dtype=torch.bfloat16, device=dev())
    pre = torch.zeros(pM, pN, dtype=torch.bfloat16, device=dev())
    ops.gemm(sa, Bh, planes[0][:M, :N], b_t=b_t, bias=bias, act=1, pre=pre[:M, :N], out2=copy[:M, :N], out_lo=planes[0].numel())
    torch.cuda.synchronize()
    _lib.check(_lib.lib().afft_set_gemm_variant(0))
    exact = _act(1, (A.double() @ Bm.half().double()).float() + bias.cpu(), None)
    assert rel_l2(ref.cpu(), exact) < 1e-5
    hi, lo = planes[0][:M, :N], planes[1][:M, :N]
    assert torch.equal(hi, ref.half())
    assert torch.equal(lo, (ref - ref.half().float()).half())
    assert torch.equal(copy[:M, :N], ref.bfloat16())
    assert float((hi.float() + lo.float() - ref).abs().max()) <= float(ref.abs().max()) * 2.0 ** -21
    assert float(planes[:, M:].abs().max() if pM > M else 0.0) == 0.0 and float(planes[:, :, N:].abs().max() if pN > N else 0.0) == 0.0
    # ... and the planes are the A operand of the next two-pass GEMM
    W2 = rnd(N, 192, seed=43)
    sp = ops.Split.__new__(ops.Split)
    sp.planes, sp.rows, sp.cols, sp.f16 = planes, M, N, True
    y = torch.zeros(M, 192, device=dev())
    W2h = torch.zeros(pN, 192, dtype=torch.float16, device=dev())
    W2h[:N] = W2.half().to(dev())
    ops.gemm(sp, W2h, y)
    torch.cuda.synchronize()
    assert rel_l2(y.cpu(), (ref.cpu().double() @ W2.half().double()).float()) < 1e-5


def _e4m3_decode(b):
    """uint8 tensor of OCP e4m3fn codes -> float32 values"""
    b = b.to(torch.int32)
    s, e, m = b >> 7, (b >> 3) & 15, b & 7
    v = torch.where(e == 0, m.float() / 8.0 * 2.0 ** -6, (1.0 + m.float() / 8.0) * torch.pow(2.0, (e - 7).float()))
    return torch.where(s == 1, -v, v)


@pytest.mark.parametrize("M,N,K", [(5120, 2048, 2048), (3000, 4096, 1024), (4096, 4096, 128), (5120, 2048, 256)])      # the last: the steady-state kernel at its shortest (two fp16 pairs, ONE fp8 pair = the frozen-offset pair)
def test_gemm_fp16_hi_pass_plus_fp8_lo_pass(M, N, K):
    """afft_gemm_t.split3 = 3: first pass A_hi W on the fp16 MFMA, second pass A_lo W on the block-scaled fp8 MFMA over e4m3 byte planes
    (a8 = e4m3(2^11 (a - hi)), b8 = e4m3(2^8 w), constant scales).  Against float64 on the SAME quantised operands the result is
    fp32-grade (the kernel computes what it says); against the exact product the error is the weight's fp16 rounding, like the
    two-pass fp16 GEMM's (the lo term carries ~2^-12 of the product: its 2^-4 operand rounding vanishes); the plane output feeds the
    next such GEMM."""
    from afft_amd import _lib, ops
    assert _lib.lib().afft_gemm_lo8_ok(M, N, K) == 1
    A, W = rnd(M, K, seed=51), rnd(N, K, seed=52, scale=0.05)
    hi = torch.zeros(M, K, dtype=torch.float16, device=dev())
    a8 = torch.zeros(M, K, dtype=torch.uint8, device=dev())
    ops.quant_e4m3(A.to(dev()), 2048.0, a8, hi=hi)
    w16 = W.half().to(dev())
    w8 = torch.zeros(N, K, dtype=torch.uint8, device=dev())
    ops.quant_e4m3(W.to(dev()), 256.0, w8)
    torch.cuda.synchronize()
    assert torch.equal(hi.cpu(), A.half())
    lo_ref = (A - A.half().float()) * 2048.0
    assert rel_l2(_e4m3_decode(a8.cpu()), lo_ref) < 0.05 and rel_l2(_e4m3_decode(w8.cpu()), W * 256.0) < 0.05      # 3 mantissa bits
    bias = rnd(N, seed=3).to(dev())
    out = torch.zeros(M, N, device=dev())
    ops.gemm(hi, w16, out, b_t=True, bias=bias, a8=a8, b8=w8)
    torch.cuda.synchronize()
    quant = (hi.cpu().double() @ w16.cpu().double().t()
             + (_e4m3_decode(a8.cpu()).double() @ _e4m3_decode(w8.cpu()).double().t()) * 2.0 ** -19).float() + bias.cpu()
    exact = (A.double() @ W.double().t()).float() + bias.cpu()
    assert rel_l2(out.cpu(), quant) < 2e-5, rel_l2(out.cpu(), quant)
    e = rel_l2(out.cpu(), exact)
    two_pass = torch.zeros(M, N, device=dev())
    ops.gemm(ops.Split(A.to(dev()), f16=True), w16, two_pass, b_t=True, bias=bias)
    torch.cuda.synchronize()
    e2 = rel_l2(two_pass.cpu(), exact)
    print(f"[{M}x{N}x{K}] vs exact: fp16 + fp8 lo pass {e:.2e}, two fp16 passes {e2:.2e}")
    assert e < 1.15 * e2 + 1e-6 and e < 4e-4
    # plane output: hi fp16 + e4m3 lo bytes of the result
    oh = torch.zeros(M, N, dtype=torch.float16, device=dev())
    o8 = torch.zeros(M, N, dtype=torch.uint8, device=dev())
    ops.gemm(hi, w16, oh, b_t=True, bias=bias, a8=a8, b8=w8, out_lo8=o8)
    torch.cuda.synchronize()
    assert torch.equal(oh, out.half())
    lo = (out - out.half().float()).cpu() * 2048.0
    assert rel_l2(_e4m3_decode(o8.cpu()), lo) < 0.05


@pytest.mark.parametrize("rows,d", [(37, 64), (300, 1024), (130, 2048)])
def test_layernorm_fwd_split_planes(rows, d):
    """afft_layernorm_fwd_split: the planes and the bf16 copy are bitwise the splits / the rounding of the fp32 LayerNorm output"""
    from afft_amd import ops
    x, w, b = rnd(rows, d, seed=1).to(dev()), rnd(d, seed=2).to(dev()), rnd(d, seed=3).to(dev())
    y = torch.empty(rows, d, device=dev())
    m0, r0 = torch.empty(rows, device=dev()), torch.empty(rows, device=dev())
    ops.layernorm_fwd(x, w, b, 1e-6, y, m0, r0)
    pr = (rows + 63) // 64 * 64
    planes = torch.zeros(2, pr, d, dtype=torch.float16, device=dev())
    copy = torch.zeros(pr, d, dtype=torch.bfloat16, device=dev())
    m1, r1 = torch.empty(rows, device=dev()), torch.empty(rows, device=dev())
    ops.layernorm_fwd_split(x, w, b, 1e-6, planes[0][:rows], planes[0].numel(), copy[:rows], m1, r1)
    torch.cuda.synchronize()
    assert torch.equal(planes[0][:rows], y.half()) and torch.equal(planes[1][:rows], (y - y.half().float()).half())
    assert torch.equal(copy[:rows], y.bfloat16()) and torch.equal(m0, m1) and torch.equal(r0, r1)
    ops.layernorm_fwd_split(x, None, None, 1e-6, planes[0][:rows], planes[0].numel(), None)     # no affine, no copy
    torch.cuda.synchronize()
    assert torch.isfinite(planes.float()).all()
    # the lo part as an e4m3 byte plane (the fp8 lo pass)
    lo8 = torch.zeros(pr, d, dtype=torch.uint8, device=dev())
    ops.layernorm_fwd_split(x, w, b, 1e-6, planes[0][:rows], 0, copy[:rows], y_lo8=lo8[:rows])
    torch.cuda.synchronize()
    assert torch.equal(planes[0][:rows], y.half()) and torch.equal(copy[:rows], y.bfloat16())
    assert rel_l2(_e4m3_decode(lo8[:rows].cpu()), ((y - y.half().float()) * 2048.0).cpu()) < 0.05


ATTN_SPLIT_CASES = [(13, 5, 4, 64, 0), (64, 5, 4, 512, 0), (9, 6, 2, 256, 1), (6, 16, 4, 128, 2), (3, 32, 2, 512, 2), (3, 64, 2, 256, 3),
                    (2, 40, 2, 64, 3)]


@pytest.mark.parametrize("nseq,L,H,hd,mask", ATTN_SPLIT_CASES)
def test_attention_fwd_split(nseq, L, H, hd, mask):
    """fp16x2 forward attention on hi + lo planes (three fp16 MFMA products per matrix product): fp32-grade against float64 --
    no 2^-11 operand rounding -- with outputs as planes + the bf16 copy; packed frames (L = 5, 6), one / two / four row tiles,
    whole and chunked head dimensions, every mask."""
    from afft_amd import ops
    from oracle import afft_oracle as O
    d = H * hd
    R = nseq * L
    qkv = rnd(R, 3 * d, seed=1)
    sp = ops.Split(qkv.to(dev()), f16=True)
    hi = sp.planes[0]
    in_lo = hi.numel()
    q, k, v = hi[:R, :d], hi[:R, d:2 * d], hi[:R, 2 * d:3 * d]
    pr = (R + 63) // 64 * 64
    oplanes = torch.zeros(2, pr, d, dtype=torch.float16, device=dev())
    ocopy = torch.zeros(pr, d, dtype=torch.bfloat16, device=dev())
    probs = torch.empty(nseq, H, L, L, device=dev())
    scale = hd ** -0.5
    period = L // 4 if mask == 3 else 0
    ops.attention_fwd_split(q, k, v, in_lo, nseq, L, H, hd, scale, mask, oplanes[0][:R], oplanes[0].numel(), ocopy[:R], probs,
                            mask_period=period)
    torch.cuda.synchronize()
    t = qkv.double().view(nseq, L, 3, H, hd).permute(2, 0, 3, 1, 4)
    if mask == 3:
        m = O.make_mask("causal", period, torch.float64).repeat(4, 4)
    else:
        m = O.make_mask(["none", "diag", "causal"][mask], L, torch.float64)
    o_ref, p_ref = O._softmax_attend(t[0], t[1], t[2], scale, m)
    o_ref = o_ref.reshape(R, d).float()
    got = (oplanes[0][:R].float() + oplanes[1][:R].float()).cpu()
    assert rel_l2(got, o_ref) < 2e-5, rel_l2(got, o_ref)
    assert rel_l2(probs.cpu(), p_ref.float()) < 2e-5
    assert rel_l2(ocopy[:R].float().cpu(), o_ref) < 4e-3 and rel_l2(oplanes[0][:R].float().cpu(), o_ref) < 5e-4
    if mask:
        assert float(probs.cpu()[..., torch.isinf(m)].abs().max()) == 0.0
    # with dropout the kernel draws the masks of the bf16 kernel (same key, same index): compare with it on bf16-exact inputs
    qb = bfr(qkv)
    spb = ops.Split(qb.to(dev()), f16=True)      # bf16 values are exact in hi + lo
    hb = spb.planes[0]
    ops.attention_fwd_split(hb[:R, :d], hb[:R, d:2 * d], hb[:R, 2 * d:3 * d], hb.numel(), nseq, L, H, hd, scale, mask, oplanes[0][:R],
                            oplanes[0].numel(), None, probs, drop_p=0.3, drop_key=777, mask_period=period)
    g = qb.to(torch.bfloat16).to(dev())
    ob = torch.empty(R, d, dtype=torch.bfloat16, device=dev())
    pb = torch.empty(nseq, H, L, L, device=dev())
    ops.attention_fwd(g[:, :d], g[:, d:2 * d], g[:, 2 * d:], nseq, L, H, hd, scale, mask, ob, pb, drop_p=0.3, drop_key=777, mask_period=period)
    torch.cuda.synchronize()
    assert rel_l2(probs.cpu(), pb.cpu()) < 1e-5
    assert rel_l2((oplanes[0][:R].float() + oplanes[1][:R].float()).cpu(), ob.float().cpu()) < 1.5e-2


def test_sgd_kernels_keep_the_fp16_image():
    from afft_amd import ops
    n = 64 * 1000
    p, g, buf = rnd(n, seed=1).to(dev()), rnd(n, seed=2).to(dev()), rnd(n, seed=3).to(dev())
    p16, h16 = torch.zeros(n, dtype=torch.bfloat16, device=dev()), torch.zeros(n, dtype=torch.float16, device=dev())
    ops.sgd_nesterov(p, g, buf, 1e-2, 0.9, 1e-4, 1.0, 0, p_bf16=p16, p_f16=h16)
    torch.cuda.synchronize()
    assert torch.equal(h16, p.half()) and torch.equal(p16, p.bfloat16())
    runs = torch.tensor([[0, 4096], [8192, 1000]], dtype=torch.int64, device=dev())
    h16.zero_()
    ops.sgd_nesterov_runs(p, g, buf, runs, 1e-2, 0.9, 1e-4, 1.0, 0, p_bf16=p16, p_f16=h16)
    torch.cuda.synchronize()
    assert torch.equal(h16[:4096], p[:4096].half()) and torch.equal(h16[8192:9192], p[8192:9192].half())
    assert float(h16[4096:8192].abs().max()) == 0.0


@pytest.mark.parametrize("rows,d,dt", [(37, 64, "f32"), (300, 1024, "bf16"), (130, 2048, "f32"), (5, 128, "bf16")])
def test_layernorm_fwd_bwd(rows, d, dt):
    from afft_amd import ops
    ydt = torch.float32 if dt == "f32" else torch.bfloat16
    x = rnd(rows, d, seed=1, scale=2.0) + 0.3
    w = rnd(d, seed=2) * 0.2 + 1.0
    b = rnd(d, seed=3) * 0.1
    eps = 1e-6
    xg = x.to(dev())
    y = torch.empty(rows, d, dtype=ydt, device=dev())
    mean = torch.empty(rows, device=dev())
    rstd = torch.empty(rows, device=dev())
    ops.layernorm_fwd(xg, w.to(dev()), b.to(dev()), eps, y, mean, rstd)
    xr = x.clone().double().requires_grad_(True)
    wr = w.clone().double().requires_grad_(True)
    br = b.clone().double().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (d,), wr, br, eps)
    assert rel_l2(y.float().cpu(), yr.float()) < (2e-5 if dt == "f32" else 5e-3)
    dy = rnd(rows, d, seed=4)
    if dt == "bf16":
        dy = bfr(dy)
    dx_in = rnd(rows, d, seed=5)
    yr.backward(dy.double())
    dx = torch.empty(rows, d, device=dev())
    dxb = torch.empty(rows, d, dtype=torch.bfloat16, device=dev())
    dw = torch.full((d,), 0.5, device=dev())
    db = torch.full((d,), -0.25, device=dev())
    ops.layernorm_bwd(dy.to(ydt).to(dev()), xg, w.to(dev()), mean, rstd, dx, dx_in=dx_in.to(dev()), dx_bf16=dxb, dw=dw, db=db)
    torch.cuda.synchronize()
    assert rel_l2(dx.cpu(), xr.grad.float() + dx_in) < 3e-5
    assert rel_l2(dxb.float().cpu(), xr.grad.float() + dx_in) < 5e-3
    assert rel_l2(dw.cpu() - 0.5, wr.grad.float()) < 3e-5
    assert rel_l2(db.cpu() + 0.25, br.grad.float()) < 3e-5
    # accumulate = False overwrites whatever the gradient buffers held
    ops.layernorm_bwd(dy.to(ydt).to(dev()), xg, w.to(dev()), mean, rstd, dx, dx_in=dx_in.to(dev()), dw=dw, db=db,
                      accumulate=False)
    assert rel_l2(dw.cpu(), wr.grad.float()) < 3e-5 and rel_l2(db.cpu(), br.grad.float()) < 3e-5


def test_layernorm_strided_token0_and_no_affine():
    from afft_amd import ops
    BT, S, d = 12, 5, 64
    X = rnd(BT * S, d, seed=9).to(dev())
    y = torch.empty(BT, d, device=dev())
    ops.layernorm_fwd(X.view(BT, S * d)[:, :d], None, None, 1e-6, y)
    ref = torch.nn.functional.layer_norm(X.cpu().view(BT, S, d)[:, 0], (d,), None, None, 1e-6)
    assert rel_l2(y.cpu(), ref) < 2e-5


ATTN_CASES = [(7, 5, 4, 16, 0, "f32"), (6, 5, 4, 64, 1, "f32"), (3, 16, 2, 64, 2, "f32"), (2, 32, 2, 32, 2, "f32"),
              (9, 5, 4, 256, 0, "bf16"), (4, 16, 4, 512, 2, "bf16"), (3, 6, 4, 128, 1, "bf16"), (5, 1, 2, 32, 0, "f32"),
              # bf16 with hd % 64 == 0 takes the MFMA path: packed frames (G=3, ragged last group), G=2, NT=2, T=10
              (7, 5, 4, 512, 0, "bf16"), (64, 5, 4, 64, 1, "bf16"), (3, 8, 2, 64, 0, "bf16"), (2, 32, 2, 64, 2, "bf16"),
              (5, 10, 4, 64, 2, "bf16"), (3, 17, 2, 128, 2, "bf16"), (4, 16, 2, 16, 2, "bf16"),
              # T-SA-Fuser sequences: L = M*T > 32 tokens, mask 3 = the causal T x T mask tiled over the modalities
              # (period T = L / 4 here); plain / causal long sequences too
              (2, 40, 2, 32, 3, "f32"), (3, 64, 4, 64, 3, "bf16"), (2, 80, 2, 64, 2, "f32"), (1, 128, 2, 32, 0, "bf16"),
              (2, 20, 2, 64, 3, "bf16"), (1, 96, 1, 16, 3, "f32"),
              # MFMA path with 4 row tiles (L <= 64) and the head dimension staged in chunks (hd = 512: 2 x 256 forward and
              # backward; hd = 1024: 4 x 256), block-causal and causal; L = 40 (ragged last tile)
              (3, 64, 2, 512, 3, "bf16"), (2, 64, 1, 1024, 2, "bf16"), (5, 40, 2, 256, 3, "bf16"), (2, 48, 4, 128, 0, "bf16"),
              (3, 32, 2, 1024, 2, "bf16")]


@pytest.mark.parametrize("nseq,L,H,hd,mask,dt", ATTN_CASES)
def test_attention_fwd_bwd(nseq, L, H, hd, mask, dt):
    from afft_amd import ops
    from oracle import afft_oracle as O
    tdt = torch.float32 if dt == "f32" else torch.bfloat16
    d = H * hd
    qkv = rnd(nseq * L, 3 * d, seed=1)
    dout = rnd(nseq * L, d, seed=2)
    if dt == "bf16":
        qkv, dout = bfr(qkv), bfr(dout)
    g = qkv.to(tdt).to(dev())
    q, k, v = g[:, :d], g[:, d:2 * d], g[:, 2 * d:]
    out = torch.empty(nseq * L, d, dtype=tdt, device=dev())
    probs = torch.empty(nseq, H, L, L, device=dev())
    scale = hd ** -0.5
    period = L // 4 if mask == 3 else 0
    ops.attention_fwd(q, k, v, nseq, L, H, hd, scale, mask, out, probs, mask_period=period)
    qr = qkv.clone().double().requires_grad_(True)
    t = qr.view(nseq, L, 3, H, hd).permute(2, 0, 3, 1, 4)
    if mask == 3:
        m = O.make_mask("causal", period, torch.float64).repeat(4, 4)
    else:
        m = O.make_mask(["none", "diag", "causal"][mask], L, torch.float64)
    o_ref, p_ref = O._softmax_attend(t[0], t[1], t[2], scale, m)
    tol = 2e-5 if dt == "f32" else 1e-2
    assert rel_l2(out.float().cpu(), o_ref.reshape(nseq * L, d).float()) < tol
    assert rel_l2(probs.cpu(), p_ref.float()) < (2e-5 if dt == "f32" else 2e-3)
    if mask:  # masked probabilities are exactly zero
        pm = probs.cpu()
        idx = torch.isinf(m)
        assert float(pm[..., idx].abs().max()) == 0.0
    o_ref.reshape(nseq * L, d).backward(dout.double())
    dg = torch.zeros(nseq * L, 3 * d, dtype=tdt, device=dev())
    ops.attention_bwd(dout.to(tdt).to(dev()), q, k, v, probs, nseq, L, H, hd, scale, dg[:, :d], dg[:, d:2 * d], dg[:, 2 * d:])
    torch.cuda.synchronize()
    assert rel_l2(dg.float().cpu(), qr.grad.float()) < (3e-5 if dt == "f32" else 2e-2)


def test_attention_dropout_mfma_matches_generic():
    """The bf16 MFMA kernels and the generic fp32 kernels derive the dropout mask from the same (key, index):
    with the same key they must agree (forward output and all three gradients)."""
    from afft_amd import ops
    nseq, L, H, hd, mask = 8, 5, 4, 64, 0
    d = H * hd
    scale = hd ** -0.5
    qkv = bfr(rnd(nseq * L, 3 * d, seed=11))
    dout = bfr(rnd(nseq * L, d, seed=12))
    res = {}
    for tdt in (torch.float32, torch.bfloat16):
        g = qkv.to(tdt).to(dev())
        q, k, v = g[:, :d], g[:, d:2 * d], g[:, 2 * d:]
        out = torch.empty(nseq * L, d, dtype=tdt, device=dev())
        probs = torch.empty(nseq, H, L, L, device=dev())
        ops.attention_fwd(q, k, v, nseq, L, H, hd, scale, mask, out, probs, drop_p=0.3, drop_key=12345)
        dg = torch.zeros(nseq * L, 3 * d, dtype=tdt, device=dev())
        ops.attention_bwd(dout.to(tdt).to(dev()), q, k, v, probs, nseq, L, H, hd, scale, dg[:, :d], dg[:, d:2 * d],
                          dg[:, 2 * d:], drop_p=0.3, drop_key=12345)
        torch.cuda.synchronize()
        res[tdt] = (out.float().cpu(), probs.cpu(), dg.float().cpu())
    o32, p32, g32 = res[torch.float32]
    o16, p16, g16 = res[torch.bfloat16]
    assert rel_l2(p16, p32) < 2e-3          # pre-dropout probabilities
    assert rel_l2(o16, o32) < 1.5e-2
    assert rel_l2(g16, g32) < 3e-2
    # dropout really happened: output differs from the p=0 output
    g = qkv.to(dev())
    out0 = torch.empty(nseq * L, d, device=dev())
    ops.attention_fwd(g[:, :d], g[:, d:2 * d], g[:, 2 * d:], nseq, L, H, hd, scale, mask, out0, None)
    assert rel_l2(out0.cpu(), o32) > 0.1


@pytest.mark.parametrize("rows,C,soft", [(37, 3806, False), (20, 11, False), (33, 3806, True), (16, 7, True)])
def test_softmax_ce(rows, C, soft):
    from afft_amd import ops
    ld = ((C + 63) // 64) * 64
    logits = rnd(rows, C, seed=1, scale=3.0)
    buf = torch.zeros(rows, ld, device=dev())
    buf[:, :C] = logits.to(dev())
    lg = buf[:, :C]
    loss_sum = torch.zeros(1, device=dev())
    row_loss = torch.empty(rows, device=dev())
    dl = torch.full((rows, ld), 7.0, dtype=torch.bfloat16, device=dev())
    lr = logits.clone().double().requires_grad_(True)
    gscale = 1.0 / rows
    if not soft:
        g = torch.Generator().manual_seed(3)
        labels = torch.randint(0, C, (rows,), generator=g)
        labels[::5] = -1
        ops.softmax_ce(lg, C, labels=labels.to(dev()), gscale=gscale, loss_sum=loss_sum, dlogits=dl, row_loss=row_loss)
        ref = torch.nn.functional.cross_entropy(lr, labels, ignore_index=-1, reduction="none")
        (ref.sum() * gscale).backward()
    else:
        t = torch.softmax(rnd(rows, C, seed=4), -1) * 0.6
        t[:, 1] += 0.4
        keep = torch.ones(rows, dtype=torch.uint8)
        keep[::4] = 0
        ops.softmax_ce(lg, C, soft=t.to(dev()), keep=keep.to(dev()), gscale=gscale, loss_sum=loss_sum, dlogits=dl, row_loss=row_loss)
        ref = torch.nn.functional.cross_entropy(lr, t.double(), reduction="none") * keep.double()
        (ref.sum() * gscale).backward()
    torch.cuda.synchronize()
    assert rel_l2(row_loss.cpu(), ref.float()) < 1e-5
    assert abs(float(loss_sum) - float(ref.sum())) < 1e-4 * max(1.0, float(ref.sum()))
    assert rel_l2(dl[:, :C].float().cpu(), lr.grad.float()) < 5e-3
    assert float(dl[:, C:].float().abs().max()) == 0.0


def test_softmax_ce_out_of_range_label_poisons_the_row():
    """a label >= C (torch: device-side assert) reads nothing out of bounds; that row's loss and gradient are NaN, the other
    rows are untouched"""
    from afft_amd import ops
    rows, C = 6, 11
    logits = rnd(rows, C, seed=1).to(dev())
    labels = torch.tensor([1, C, 3, -1, 10, C + 100]).to(dev())
    row_loss = torch.empty(rows, device=dev())
    dl = torch.zeros(rows, C, device=dev())
    ops.softmax_ce(logits, C, labels=labels, dlogits=dl, row_loss=row_loss)
    torch.cuda.synchronize()
    rl, g = row_loss.cpu(), dl.cpu()
    assert torch.isnan(rl[[1, 5]]).all() and torch.isnan(g[[1, 5]]).all()
    assert torch.isfinite(rl[[0, 2, 3, 4]]).all() and torch.isfinite(g[[0, 2, 3, 4]]).all() and float(rl[3]) == 0.0
    ref = torch.nn.functional.cross_entropy(logits.cpu()[[0, 2, 4]], labels.cpu()[[0, 2, 4]], reduction="none")
    assert rel_l2(rl[[0, 2, 4]], ref) < 1e-5


def _pack_ref(w):
    """torch restatement of afft_pack_weight (include/afft_hip.h): [rows, cols] -> [rows / 16][cols / 32][lane = (r & 15) + 16 ((c >> 3) & 3)][8]"""
    R, Cc = w.shape
    return w.reshape(R // 16, 16, Cc // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().reshape(-1)


BD_CASES = [
    # name, variant (7: 256x256 tiles, B row-major; 8: 160x256; 9 / 10: the same with a fragment-packed B), M, N, K, epilogue
    ("bd160_one_ktile", 8, 160, 256, 64, dict(out_f32=True)),
    ("bd160_short_k", 8, 320, 512, 192, dict(bias=True, out_f32=True)),            # fewer K-tiles than the look-ahead
    ("bd160_tails", 8, 4100, 2064, 1024, dict(bias=True, residual=True, out_f32=True, alpha=0.5)),
    ("bd160_path_proj", 8, 5120, 2048, 2048, dict(bias=True, residual=True, out_f32=True)),
    ("bd160_gelu", 8, 1000, 768, 448, dict(bias=True, act=1, pre=True)),
    ("bd256_tails", 7, 4100, 2064, 1024, dict(bias=True, residual=True, out_f32=True)),
    ("bd256_gelu", 7, 600, 512, 320, dict(bias=True, act=2, pre=True)),
    ("bd160_packed_one_ktile", 10, 160, 256, 64, dict(out_f32=True)),
    ("bd160_packed_tails", 10, 4100, 2064, 1024, dict(bias=True, residual=True, out_f32=True)),
    ("bd160_packed_path_fc2", 10, 5120, 2048, 8192, dict(bias=True, residual=True, out_f32=True)),
    ("bd160_packed_gelu", 10, 1000, 768, 448, dict(bias=True, act=1, pre=True)),
    ("bd256_packed", 9, 1000, 1040, 704, dict(bias=True, out_f32=True)),
    ("auto_packed_path_proj", 0, 5120, 2048, 2048, dict(bias=True, residual=True, out_f32=True)),   # b_packed given: the dispatcher's choice
]


@pytest.mark.parametrize("case", BD_CASES, ids=[c[0] for c in BD_CASES])
def test_gemm_b_direct(case):
    """csrc/gemm_bd.hip: B operand global -> register (row-major or fragment-packed), A through LDS, 160- and 256-row tiles,
    NT layout: against fp64 math, row / column tails, K shorter than the look-ahead, every epilogue stage the forward GEMMs
    use; repeated launches give the same bits."""
    from afft_amd import _lib, ops
    name, variant, M, N, K, ep = case
    A, W = bfr(rnd(M, K, seed=11)), bfr(rnd(N, K, seed=12))
    a = A.to(torch.bfloat16).to(dev())
    w = W.to(torch.bfloat16).to(dev())
    packed = None
    if variant in (9, 10) or variant == 0:
        Np = (N + 15) // 16 * 16
        wp = torch.zeros(Np, K, dtype=torch.float32, device=dev())
        wp[:N] = W.to(dev())
        packed = torch.empty(Np * K, dtype=torch.bfloat16, device=dev())
        ops.pack_weight(wp, packed)
        assert torch.equal(packed, _pack_ref(wp.to(torch.bfloat16)))
    bias = rnd(N, seed=13).to(dev()) if ep.get("bias") else None
    res = rnd(M, N, seed=14).to(dev()) if ep.get("residual") else None
    act = ep.get("act", 0)
    out = torch.zeros(M, N, dtype=torch.float32 if ep.get("out_f32") else torch.bfloat16, device=dev())
    pre = torch.zeros(M, N, dtype=torch.bfloat16, device=dev()) if ep.get("pre") else None
    alpha = ep.get("alpha", 1.0)
    _lib.check(_lib.lib().afft_set_gemm_variant(variant))
    try:
        def run():
            if variant in (9, 10):      # forced packed variants read the packed image through B
                ops.gemm(a, packed.view(-1, K)[:N], out, b_t=True, bias=bias, act=act, pre=pre, residual=res, alpha=alpha)
            else:
                ops.gemm(a, w, out, b_t=True, bias=bias, act=act, pre=pre, residual=res, alpha=alpha,
                         b_packed=packed if (variant == 0 and N % 16 == 0) else None)
        run()
        torch.cuda.synchronize()
        first = out.clone()
        ref = alpha * (A.double() @ W.double().t()).float()
        if bias is not None:
            ref = ref + bias.cpu()
        pre_ref = ref.clone()
        ref = _act(act, ref, None)
        if res is not None:
            ref = ref + res.cpu()
        assert rel_l2(out.float().cpu(), ref) < (2e-3 if ep.get("out_f32") else 1e-2), name
        if pre is not None:
            assert rel_l2(pre.float().cpu(), pre_ref) < 1e-2
        for _ in range(5):
            run()
            assert torch.equal(out, first)
        if variant == 0:       # the dispatcher really took the B-direct kernel for this shape
            assert _lib.lib().afft_gemm_variant_for(M, N, K, 0, 0) in (1, 3)       # ... which the row-major query does not know about
    finally:
        _lib.check(_lib.lib().afft_set_gemm_variant(0))


def test_fused_sgd_epilogue_keeps_the_packed_image_fresh():
    """afft_sgd_fused_t.p_pk16: the weight-gradient epilogue that updates a weight also rewrites its fragment-packed image:
    == afft_pack_weight of the updated fp32 weight (bf16 rounding of the same values), 256x256 and 128x128 tile epilogues"""
    from afft_amd import _lib, ops
    for variant, M, N, K in ((3, 2048, 1024, 640), (1, 384, 256, 320)):
        _lib.check(_lib.lib().afft_set_gemm_variant(variant))
        a = bfr(rnd(K, M, seed=51)).to(torch.bfloat16).to(dev())
        b = bfr(rnd(K, N, seed=52)).to(torch.bfloat16).to(dev())
        p, m = rnd(M, N, seed=53).to(dev()), rnd(M, N, seed=54).to(dev())
        p16 = torch.zeros(M, N, dtype=torch.bfloat16, device=dev())
        pk = torch.zeros(M * N, dtype=torch.bfloat16, device=dev())
        d = _lib.SgdFused()
        d.p, d.buf, d.p_bf16, d.p_pk16 = p.data_ptr(), m.data_ptr(), p16.data_ptr(), pk.data_ptr()
        d.lr, d.mom, d.wd, d.gscale, d.first_step = 0.05, 0.9, 1e-3, 0.5, 0
        gout = torch.empty(M, N, device=dev())
        ops.gemm(a, b, gout, a_t=True, sgd=d)
        torch.cuda.synchronize()
        _lib.check(_lib.lib().afft_set_gemm_variant(0))
        assert torch.equal(p16, p.to(torch.bfloat16))
        want = torch.empty(M * N, dtype=torch.bfloat16, device=dev())
        ops.pack_weight(p, want)
        assert torch.equal(pk, want) and torch.equal(pk, _pack_ref(p16))


@pytest.mark.parametrize("case", [("pp", 3, 1, 2048, 1024, 640), ("pp_tail", 3, 1, 2048, 1100, 640), ("small", 1, 1, 384, 256, 320),
                                  ("small_splitk", 1, 2, 512, 512, 1024)],
                         ids=lambda c: c[0])
def test_gemm_fused_sgd_epilogue_equals_gemm_then_update(case):
    """afft_gemm_t.sgd: the Nesterov update applied in the weight-gradient GEMM's epilogue (parameters, momentum and bf16 image
    written, gradient never stored) == the same GEMM storing the gradient followed by afft_sgd_nesterov, bit for bit: 256x256
    and 128x128 tiles, column tails (scalar epilogue path), split-K (the last-arriving slice runs the fused epilogue)."""
    from afft_amd import _lib, ops
    name, variant, splitk, M, N, K = case
    _lib.check(_lib.lib().afft_set_gemm_variant(variant))
    _lib.check(_lib.lib().afft_set_gemm_splitk(splitk))
    ld = (N + 63) // 64 * 64
    a = bfr(rnd(K, M, seed=41)).to(torch.bfloat16).to(dev())
    bb = torch.zeros(K, ld, dtype=torch.bfloat16, device=dev())
    bb[:, :N] = bfr(rnd(K, N, seed=42)).to(torch.bfloat16).to(dev())
    b = bb[:, :N]
    p0, m0 = rnd(M, ld, seed=43).to(dev()), rnd(M, ld, seed=44).to(dev())
    lr, mom, wd, gs = 0.05, 0.9, 1e-3, 0.5
    # reference: store the gradient, then the stand-alone update kernel
    g = torch.zeros(M, ld, device=dev())
    ops.gemm(a, b, g[:, :N], a_t=True)
    p_ref, m_ref, p16_ref = p0.clone(), m0.clone(), torch.zeros(M, ld, dtype=torch.bfloat16, device=dev())
    ops.sgd_nesterov(p_ref.view(-1), g.view(-1), m_ref.view(-1), lr, mom, wd, gs, False, p_bf16=p16_ref.view(-1))
    # fused
    p, m, p16 = p0.clone(), m0.clone(), torch.zeros(M, ld, dtype=torch.bfloat16, device=dev())
    d = _lib.SgdFused()
    d.p, d.buf, d.p_bf16, d.lr, d.mom, d.wd, d.gscale, d.first_step = p.data_ptr(), m.data_ptr(), p16.data_ptr(), lr, mom, wd, gs, 0
    gout = torch.full((M, ld), 7.0, device=dev())
    ops.gemm(a, b, gout[:, :N], a_t=True, sgd=d)
    torch.cuda.synchronize()
    _lib.check(_lib.lib().afft_set_gemm_splitk(1))
    _lib.check(_lib.lib().afft_set_gemm_variant(0))
    assert torch.equal(p[:, :N], p_ref[:, :N]) and torch.equal(m[:, :N], m_ref[:, :N]) and torch.equal(p16[:, :N], p16_ref[:, :N])
    assert torch.equal(p[:, N:], p0[:, N:]) and torch.equal(m[:, N:], m0[:, N:])      # padding columns untouched by the fused form
    assert float(gout.min()) == 7.0 and float(gout.max()) == 7.0                      # the gradient is never stored
    with pytest.raises(RuntimeError, match="fused update"):
        ops.gemm(a, b, gout[:, :N], a_t=True, sgd=d, accumulate=True)


@pytest.mark.parametrize("variant", [1, 3])
def test_epilogue_activation_accuracy(variant):
    """The activation math of the bf16-operand GEMM epilogues (erf by Abramowitz & Stegun 7.1.26, tanh by exp, on the hardware exp
    and reciprocal; include/afft_hip.h, afft_gemm_t.act) against float64 erf / tanh on a dense grid of bf16-representable
    arguments in [-9, 9]: the product is made exact (A = grid values, B = identity), so what is compared is the activation alone.
    Absolute error <= 1e-6 for both GELUs and the erf derivative, <= 4e-6 for the derivative of gelu_new."""
    import math
    from afft_amd import _lib, ops
    M, K = 1024, 64
    grid = torch.linspace(-9.0, 9.0, M * K).to(torch.bfloat16)
    a = grid.view(M, K).to(dev())
    eye = torch.eye(K, dtype=torch.bfloat16, device=dev())
    x = grid.view(M, K).double()
    ones_a = torch.zeros(M, K, dtype=torch.bfloat16, device=dev())
    ones_a[:, 0] = 1.0
    ones_b = torch.zeros(K, K, dtype=torch.bfloat16, device=dev())
    ones_b[0, :] = 1.0
    phi = 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))
    u = math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)
    t = torch.tanh(u)
    ref = {ops.ACT_GELU_ERF: x * phi, ops.ACT_GELU_TANH: 0.5 * x * (1.0 + t),
           ops.ACT_DGELU_ERF: phi + x * torch.exp(-0.5 * x * x) / math.sqrt(2.0 * math.pi),
           ops.ACT_DGELU_TANH: 0.5 * (1.0 + t) + 0.5 * x * (1.0 - t * t) * math.sqrt(2.0 / math.pi) * (1.0 + 3 * 0.044715 * x * x)}
    _lib.check(_lib.lib().afft_set_gemm_variant(variant))
    try:
        for act, want in ref.items():
            out = torch.empty(M, K, dtype=torch.float32, device=dev())
            if act in (ops.ACT_DGELU_ERF, ops.ACT_DGELU_TANH):      # product = 1 everywhere, the argument comes in as aux
                ops.gemm(ones_a, ones_b, out, act=act, aux=a)
            else:
                ops.gemm(a, eye, out, act=act)
            err = float((out.double().cpu() - want).abs().max())
            assert err < (4e-6 if act == ops.ACT_DGELU_TANH else 1e-6), (act, err)
    finally:
        _lib.check(_lib.lib().afft_set_gemm_variant(0))


@pytest.mark.parametrize("variant,M,N,K,mode", [(1, 1024, 1024, 512, 4), (1, 2048, 1024, 1024, 2)])
def test_splitk_handoff_stress(variant, M, N, K, mode):
    """The split-K hand-off (slices park their partial tile with write-through stores, the slice that arrives last adds them up in
    slice order) under the worst timing: SHORT slices (2-4 K-tiles, the last store of a slice is a few hundred nanoseconds before
    the last arriver's loads), 400 launches back to back, while a second stream keeps every XCD busy with large GEMMs (as the
    data-gradient chain does beside the weight gradients of a training step).  Every launch must reproduce the first one bit for
    bit -- a partial read before it became visible would show as a difference."""
    from afft_amd import _lib, ops
    a = bfr(rnd(K, M, seed=61)).to(torch.bfloat16).to(dev())
    b = bfr(rnd(K, N, seed=62)).to(torch.bfloat16).to(dev())
    big_a = torch.randn(4096, 2048, device=dev()).to(torch.bfloat16)
    big_b = torch.randn(4096, 2048, device=dev()).to(torch.bfloat16)
    big_o = torch.empty(4096, 4096, dtype=torch.bfloat16, device=dev())
    out = torch.empty(M, N, dtype=torch.float32, device=dev())
    bad = torch.zeros((), dtype=torch.int64, device=dev())
    side = torch.cuda.Stream()
    _lib.check(_lib.lib().afft_set_gemm_variant(variant))
    _lib.check(_lib.lib().afft_set_gemm_splitk(mode))
    old_ws = ops._WS_BYTES
    ops.set_workspace_bytes(max(old_ws, 128 << 20))
    try:
        ops.gemm(a, b, out, a_t=True)
        first = out.clone()
        ref = (a.float().t().double() @ b.float().double()).float()
        assert rel_l2(first.cpu(), ref.cpu()) < 1e-5
        torch.cuda.synchronize()
        for i in range(400):
            if i % 4 == 0:
                with torch.cuda.stream(side):
                    ops.gemm(big_a, big_b, big_o, b_t=True)
            ops.gemm(a, b, out, a_t=True)
            bad += (out != first).sum()
        torch.cuda.synchronize()
        assert int(bad) == 0, f"{int(bad)} elements differed over 400 launches"
    finally:
        torch.cuda.synchronize()
        ops.set_workspace_bytes(old_ws)      # whatever the default is: later tests must not run with a smaller scratch
        _lib.check(_lib.lib().afft_set_gemm_splitk(1))
        _lib.check(_lib.lib().afft_set_gemm_variant(0))


@pytest.mark.parametrize("rows,cols,dt", [(5120, 2048, "bf16"), (1000, 520, "f32"), (333, 130, "bf16"), (77, 3806, "f32"), (5, 64, "bf16")])
def test_colsum_is_exact_enough_and_run_to_run_identical(rows, cols, dt):
    """bias-gradient column sums: no float atomics (row blocks are added up in block order by a second kernel, or one
    workgroup walks a column strip), so repeated launches give bit-identical results -- and with them the whole training step,
    every other reduction being ordered."""
    from afft_amd import _lib, ops
    src = rnd(rows, cols, seed=51)
    s = (src.to(torch.bfloat16) if dt == "bf16" else src).to(dev())
    ref = (bfr(src) if dt == "bf16" else src).double().sum(0).float()
    out = torch.full((cols,), 3.0, device=dev())
    ops.colsum(s, out, accumulate=False)            # row blocks through the stream's scratch, added up in block order
    first = out.clone()
    assert rel_l2(first.cpu(), ref) < 1e-5
    for _ in range(5):
        out.fill_(-1.0)
        ops.colsum(s, out, accumulate=False)
        assert torch.equal(out, first)
    ops.colsum(s, out, accumulate=True)
    assert rel_l2(out.cpu(), 2 * ref) < 1e-5
    assert int(ops.gemm_workspace(s.device)[:4096].view(torch.int32).abs().sum()) == 0     # the split-K counters are not touched
    # without a workspace (one workgroup per column strip) and with one too small for 128-row blocks (fewer, longer blocks)
    stream = torch.cuda.current_stream().cuda_stream
    small = torch.zeros(4096 + 3 * ((cols + 255) // 256) * 1024, dtype=torch.uint8, device=dev())
    for ws, nbytes in ((None, 0), (small, small.numel())):
        o = torch.full((cols,), 7.0, device=dev())
        _lib.check(_lib.lib().afft_colsum(s.data_ptr(), s.stride(0), 1 if dt == "bf16" else 0, rows, cols, o.data_ptr(), 0,
                                          ws.data_ptr() if ws is not None else None, nbytes, stream), "colsum")
        o2 = torch.empty_like(o)
        _lib.check(_lib.lib().afft_colsum(s.data_ptr(), s.stride(0), 1 if dt == "bf16" else 0, rows, cols, o2.data_ptr(), 0,
                                          ws.data_ptr() if ws is not None else None, nbytes, stream), "colsum")
        assert rel_l2(o.cpu(), ref) < 1e-5 and torch.equal(o, o2)
    if cols > 8:        # a column-offset view (unaligned rows): the generic one-thread-per-column form
        v = s[:, 1:cols - 2]
        o2 = torch.zeros(cols - 3, device=dev())
        ops.colsum(v, o2)
        assert rel_l2(o2.cpu(), ref[1:cols - 2]) < 1e-5


def _busy_neighbour(stream, n=6):
    """a second stream that keeps the chip busy while the reductions under test run (arrival order of workgroups changes)"""
    a = torch.randn(2048, 2048, device=dev())
    with torch.cuda.stream(stream):
        for _ in range(n):
            a = torch.tanh(a * 1.0001)
    return a


@pytest.mark.parametrize("rows,d", [(9, 64), (21, 64), (960, 2048), (1984, 2048), (2047, 1030)])
def test_mse_loss_scalar_is_bit_stable(rows, d):
    """the MSE scalar of the loss (common/runner.py:164-166) is summed without float atomics: 200 launches on the sizes of the
    goldens t2_flt (B 3 x (T-1) 3 rows of 64), t3_m5, and of the bench workload cfg2 (64 x 15 rows of 2048) / cfg5 give ONE bit
    pattern, alone and beside a second busy stream; the value agrees with float64"""
    from afft_amd import ops
    a, b = rnd(rows, d, seed=61).to(dev()), rnd(rows, d, seed=62).to(dev())
    ref = float(((a.double() - b.double()) ** 2).mean())
    side = torch.cuda.Stream()
    seen = set()
    keep = []
    for i in range(200):
        if i % 20 == 10:
            keep.append(_busy_neighbour(side))
        ls = torch.zeros(1, device=dev())
        ops.mse(a, b, 1.0, ls, None, None, lscale=1.0 / (rows * d))
        seen.add(int(ls.view(torch.int32).item()))
    torch.cuda.synchronize()
    assert len(seen) == 1, seen
    got = torch.tensor([seen.pop()], dtype=torch.int32).view(torch.float32).item()
    assert abs(got - ref) < 2e-6 * abs(ref)
    # accumulating form: a second call adds to the first
    ls = torch.zeros(1, device=dev())
    ops.mse(a, b, 1.0, ls, None, None, lscale=1.0 / (rows * d))
    ops.mse(a, b, 1.0, ls, None, None, lscale=1.0 / (rows * d))
    assert abs(float(ls) - 2 * ref) < 4e-6 * abs(ref)
    assert int(ops.gemm_workspace(a.device)[:4096].view(torch.int32).abs().sum()) == 0     # the split-K counters are not touched


@pytest.mark.parametrize("n,dt", [(1000, "f32"), (5_000_003, "f32"), (40_000_000, "bf16")])
def test_sumsq_and_row_loss_sum_are_bit_stable(n, dt):
    """gradient-norm partial sums (train.py:254-260 clip_grad_norm_) and the summed row losses of softmax_ce: ordered, so the
    clipping coefficient - and with it the parameters - are bit-reproducible"""
    from afft_amd import ops
    x = rnd(n, seed=63)
    xd = (x.to(torch.bfloat16) if dt == "bf16" else x).to(dev())
    ref = float((xd.double() ** 2).sum()) * 0.25
    side = torch.cuda.Stream()
    seen, keep = set(), []
    for i in range(100):
        if i % 20 == 10:
            keep.append(_busy_neighbour(side))
        out = torch.zeros(1, device=dev())
        ops.sumsq(xd, out, 0.25)
        seen.add(int(out.view(torch.int32).item()))
    assert len(seen) == 1, seen
    got = torch.tensor([seen.pop()], dtype=torch.int32).view(torch.float32).item()
    assert abs(got - ref) < 1e-5 * ref
    if dt == "f32" and n == 1000:
        rows, C = 1088, 3806
        lg = rnd(rows, C, seed=64, scale=2.0).to(dev())
        labels = torch.randint(0, C, (rows,), generator=torch.Generator().manual_seed(65)).to(dev())
        seen = set()
        for i in range(100):
            ls, rl = torch.zeros(1, device=dev()), torch.empty(rows, device=dev())
            ops.softmax_ce(lg, C, labels=labels, loss_sum=ls, row_loss=rl)
            seen.add(int(ls.view(torch.int32).item()))
        assert len(seen) == 1, seen
        assert abs(float(ls) - float(rl.double().sum())) < 1e-5 * float(rl.double().sum())


def test_sgd_runs_equals_sgd_over_the_same_ranges():
    """afft_sgd_nesterov_runs over a table of runs == afft_sgd_nesterov over each range, bit for bit; elements outside the runs
    are untouched"""
    from afft_amd import ops
    n = 40000
    g = torch.Generator().manual_seed(9)
    p0, gr, b0 = (torch.randn(n, generator=g).to(dev()) for _ in range(3))
    runs = [(0, 64), (128, 5000), (5184, 16384), (30016, 4)]
    ref_p, ref_b, ref_16 = p0.clone(), b0.clone(), torch.zeros(n, dtype=torch.bfloat16, device=dev())
    for s, ln in runs:
        ops.sgd_nesterov(ref_p[s:s + ln], gr[s:s + ln], ref_b[s:s + ln], 0.01, 0.9, 1e-4, 0.5, False, p_bf16=ref_16[s:s + ln])
    p, b, p16 = p0.clone(), b0.clone(), torch.zeros(n, dtype=torch.bfloat16, device=dev())
    ops.sgd_nesterov_runs(p, gr, b, torch.tensor(runs, dtype=torch.int64, device=dev()), 0.01, 0.9, 1e-4, 0.5, False, p_bf16=p16)
    torch.cuda.synchronize()
    assert torch.equal(p, ref_p) and torch.equal(b, ref_b) and torch.equal(p16, ref_16)
    assert torch.equal(p[64:128], p0[64:128]) and torch.equal(p[30020:], p0[30020:])


def test_mse_and_elementwise():
    from afft_amd import ops
    rows, d = 45, 64
    a, b = rnd(rows, d, seed=1), rnd(rows, d, seed=2)
    ls = torch.zeros(1, device=dev())
    da = torch.full((rows, d), 2.0 ** -10, device=dev())
    db = torch.zeros(rows, d, device=dev())
    gs = 1.0 / (rows * d)
    ops.mse(a.to(dev()), b.to(dev()), gs, ls, da, db)
    torch.cuda.synchronize()
    assert abs(float(ls) - float(((a - b) ** 2).sum())) < 1e-3
    assert rel_l2(da.cpu() - 2.0 ** -10, 2 * gs * (a - b)) < 1e-4
    assert rel_l2(db.cpu(), -2 * gs * (a - b)) < 1e-5

    # cast + transpose with padding
    W = rnd(70, 24, seed=3)
    w16 = torch.zeros(70, 64, dtype=torch.bfloat16, device=dev())
    wt16 = torch.zeros(24, 128, dtype=torch.bfloat16, device=dev())
    ops.cast(W.to(dev()), w16, wt16, zero_pad=True)
    torch.cuda.synchronize()
    assert torch.equal(w16[:, :24].cpu(), W.to(torch.bfloat16)) and float(w16[:, 24:].float().abs().max()) == 0
    assert torch.equal(wt16[:, :70].cpu(), W.t().to(torch.bfloat16)) and float(wt16[:, 70:].float().abs().max()) == 0

    # token assembly (SA-Fuser, models/fusion.py:338-352)
    B, T, dd, Mn = 3, 4, 64, 4
    feats = [rnd(B * T, dd, seed=10 + i) for i in range(Mn)]
    tok = rnd(T, dd, seed=20)
    emb = rnd(Mn + 1, dd, seed=21)
    X = torch.empty(B * T, Mn + 1, dd, device=dev())
    ops.assemble_tokens([f.to(dev()) for f in feats], tok.to(dev()), dd, emb.to(dev()), B * T, T, dd, X)
    ref = torch.stack([tok.repeat(B, 1)] + feats, dim=1) + emb
    assert torch.equal(X.cpu(), ref)
    X2 = torch.empty(B * T, Mn + 1, dd, device=dev())
    ops.assemble_tokens([f.to(dev()) for f in feats], tok[:1].contiguous().to(dev()), 0, None, B * T, T, dd, X2)
    assert torch.equal(X2.cpu(), torch.stack([tok[:1].expand(B * T, dd)] + feats, dim=1))

    # column sums (bias gradients)
    src = rnd(333, 130, seed=30)
    out = torch.full((130,), 2.0, device=dev())
    ops.colsum(src.to(torch.bfloat16).to(dev()), out, accumulate=True)
    assert rel_l2(out.cpu() - 2.0, bfr(src).sum(0)) < 1e-4
    ops.colsum(src.to(dev()), out, accumulate=False)
    assert rel_l2(out.cpu(), src.sum(0)) < 1e-5

    # token means of the fusers without a modality token, and their backward (a broadcast)
    xg = rnd(6 * 5, 72, seed=34)
    yg = torch.empty(6, 72, device=dev())
    ops.group_sum(xg.to(dev()), 6, 5, 72, 1.0 / 5, yg)
    assert rel_l2(yg.cpu(), xg.view(6, 5, 72).mean(1)) < 1e-6
    dxg = torch.empty(6 * 5, 72, device=dev())
    ops.group_bcast(yg, 6, 5, 72, 0.2, dxg)
    assert torch.allclose(dxg.cpu().view(6, 5, 72), (yg.cpu() * 0.2)[:, None, :].expand(6, 5, 72))

    # periodic tables
    x = rnd(B * T, dd, seed=31)
    tab = rnd(T, dd, seed=32)
    y = torch.empty(B * T, dd, device=dev())
    ops.add_rows_periodic(x.to(dev()), tab.to(dev()), T, y)
    assert torch.equal(y.cpu(), x + tab.repeat(B, 1))
    acc = torch.ones(T, dd, device=dev())
    ops.reduce_rows_periodic(x.to(dev()), T, acc)
    assert rel_l2(acc.cpu() - 1.0, x.view(B, T, dd).sum(0)) < 1e-5
    # strided source rows (token 0 of every frame), period 1 and period T; ragged row count (rows % period != 0)
    xs = rnd(B * T, 3 * dd, seed=33)
    acc1 = torch.zeros(1, dd, device=dev())
    ops.reduce_rows_periodic(xs.to(dev())[:, :dd], 1, acc1)
    assert rel_l2(acc1.cpu()[0], xs[:, :dd].sum(0)) < 1e-5
    accT = torch.zeros(T, dd, device=dev())
    ops.reduce_rows_periodic(xs.to(dev())[:, :dd], T, accT)
    assert rel_l2(accT.cpu(), xs[:, :dd].reshape(B, T, dd).sum(0)) < 1e-5
    accR = torch.zeros(T, dd, device=dev())
    ops.reduce_rows_periodic(x.to(dev())[:B * T - 3], T, accR)
    refR = torch.cat([x[:B * T - 3], torch.zeros(3, dd)]).view(B, T, dd).sum(0)
    assert rel_l2(accR.cpu(), refR) < 1e-5

    # gradient clipping by global norm, coefficient kept on the device
    gfl = rnd(5000 + 7, seed=43) * 3.0
    for dt_, tol_ in ((torch.float32, 1e-6), (torch.bfloat16, 1e-6)):
        gdev = gfl.to(dt_).to(dev())
        ss = torch.zeros(1, device=dev())
        ops.sumsq(gdev, ss, 0.25)
        coef, nrm = torch.empty(1, device=dev()), torch.empty(1, device=dev())
        ops.clip_coef(ss, 2.0, coef, nrm)
        ref_norm = float((gdev.float().cpu() * 0.5).norm())
        assert abs(float(nrm) - ref_norm) < 1e-4 * ref_norm
        assert abs(float(coef) - min(1.0, 2.0 / (ref_norm + 1e-6))) < 1e-5
    p0_c = rnd(1003, seed=44)
    pc = torch.nn.Parameter(p0_c.clone())
    optc = torch.optim.SGD([pc], lr=0.1, momentum=0.9, nesterov=True, weight_decay=1e-3)
    gcl = rnd(1003, seed=45) * 5.0
    pc.grad = gcl.clone()
    torch.nn.utils.clip_grad_norm_([pc], 1.5)
    optc.step()
    pgc, bufc = p0_c.clone().to(dev()), torch.empty(1003, device=dev())
    ssc, coefc = torch.zeros(1, device=dev()), torch.empty(1, device=dev())
    ops.sumsq(gcl.to(dev()), ssc)
    ops.clip_coef(ssc, 1.5, coefc)
    ops.sgd_nesterov(pgc, gcl.to(dev()), bufc, 0.1, 0.9, 1e-3, 1.0, True, gscale_dev=coefc)
    assert rel_l2(pgc.cpu(), pc.detach()) < 1e-6

    # Nesterov SGD vs torch.optim.SGD
    n = 1000 + 3
    p0, g0, g1 = rnd(n, seed=40), rnd(n, seed=41), rnd(n, seed=42)
    pr = torch.nn.Parameter(p0.clone())
    opt = torch.optim.SGD([pr], lr=0.1, momentum=0.9, nesterov=True, weight_decay=1e-3)
    pg, buf = p0.clone().to(dev()), torch.empty(n, device=dev())
    for step, g in enumerate((g0, g1)):
        pr.grad = g.clone()
        opt.step()
        ops.sgd_nesterov(pg, g.to(dev()), buf, 0.1, 0.9, 1e-3, 1.0, step == 0)
    assert rel_l2(pg.cpu(), pr.detach()) < 1e-6


@pytest.mark.parametrize("act", ["none", "relu", "gelu", "gate"])
@pytest.mark.parametrize("p_drop", [0.0, 0.4])
def test_linear_act_forward_backward(act, p_drop):
    """functional.LinearAct (fused activation + output dropout in the GEMM epilogue, one act_bwd kernel in backward)
    against torch autograd with the SAME dropout mask (recovered from the output)."""
    import afft_amd
    from afft_amd import dropout as D_, functional as F_, runtime as rt
    afft_amd.set_precision("fp32")
    rt.set_grad_mode("autograd")
    try:
        rows, n_in, n_out = 37, 48, 40
        x = rnd(rows, n_in, seed=51)
        W = rnd(n_out, n_in, seed=52) * 0.3
        b = rnd(n_out, seed=53) * 0.1
        aux = rnd(rows, n_out, seed=54) if act == "gate" else None
        xg, Wg, bg = (t.clone().to(dev()).requires_grad_(True) for t in (x, W, b))
        auxg = aux.clone().to(dev()).requires_grad_(True) if aux is not None else None
        D_.manual_seed(3)
        desc = D_.elementwise(p_drop) if p_drop > 0 else None
        y = F_.LinearAct.apply(xg, Wg, bg, act, auxg, desc)
        xr, Wr, br = (t.clone().double().requires_grad_(True) for t in (x, W, b))
        auxr = aux.clone().double().requires_grad_(True) if aux is not None else None
        pre = xr @ Wr.t() + br
        a = {"none": lambda t: t, "relu": torch.relu, "gelu": lambda t: torch.nn.functional.gelu(t),
             "gate": lambda t: auxr * torch.sigmoid(t)}[act](pre)
        if p_drop > 0:
            kept = (y.detach().cpu() != 0) | (a.detach().abs() < 1e-12)      # a dropped element is exactly 0
            frac = 1.0 - float(kept.float().mean())
            assert abs(frac - p_drop) < 0.08 or act == "relu"                # relu zeros are indistinguishable: only the replay matters
            ref = a * kept.double() / (1.0 - p_drop)
        else:
            ref = a
        assert rel_l2(y.detach().cpu(), ref.detach().float()) < 2e-5
        dy = rnd(rows, n_out, seed=55)
        y.backward(dy.to(dev()))
        ref.backward(dy.double())
        assert rel_l2(xg.grad.cpu(), xr.grad.float()) < 3e-5
        assert rel_l2(Wg.grad.cpu(), Wr.grad.float()) < 3e-5
        assert rel_l2(bg.grad.cpu(), br.grad.float()) < 3e-5
        if aux is not None:
            assert rel_l2(auxg.grad.cpu(), auxr.grad.float()) < 3e-5
    finally:
        rt.set_grad_mode("sink")
        afft_amd.set_precision("bf16")


def test_softmax_small_and_weighted_sum():
    from afft_amd import functional as F_, runtime as rt
    rt.set_grad_mode("autograd")
    try:
        rows, M, C = 29, 5, 130
        lg = rnd(rows, M, seed=61) * 2
        xs = [rnd(rows, C, seed=62 + i) for i in range(M)]
        lgg = lg.clone().to(dev()).requires_grad_(True)
        xsg = [t.clone().to(dev()).requires_grad_(True) for t in xs]
        w = F_.SoftmaxSmall.apply(lgg)
        out = F_.WeightedSum.apply(w, *xsg)
        lr = lg.clone().double().requires_grad_(True)
        xr = [t.clone().double().requires_grad_(True) for t in xs]
        wr = lr.softmax(-1)
        ref = sum(wr[:, i:i + 1] * xr[i] for i in range(M))
        assert rel_l2(w.detach().cpu(), wr.detach().float()) < 1e-6
        assert rel_l2(out.detach().cpu(), ref.detach().float()) < 1e-6
        dy = rnd(rows, C, seed=70)
        out.backward(dy.to(dev()))
        ref.backward(dy.double())
        assert rel_l2(lgg.grad.cpu(), lr.grad.float()) < 2e-5
        for a, b in zip(xsg, xr):
            assert rel_l2(a.grad.cpu(), b.grad.float()) < 1e-6
    finally:
        rt.set_grad_mode("sink")


@pytest.mark.parametrize("B,T,K,n_ign", [(6, 4, 11, 2), (5, 3, 7, 4), (4, 4, 9, 0), (8, 2, 3806, 3)])
def test_mixup_prologue_matches_oracle(B, T, K, n_ign):
    """common.mixup.MixUp on the GPU (plan / rows / labels kernels) against the oracle's restatement of
    common/mixup.py:119-182 (itself pinned to the reference by the t2_soft golden), incl. the <= 1 participant case."""
    from afft_amd.common.mixup import MixUp
    from oracle import afft_oracle as O
    g = torch.Generator().manual_seed(B * 100 + T)
    feats = {"rgb": torch.randn(B, T, 24, generator=g), "flow": torch.randn(B, T, 16, generator=g)}
    tgt = torch.randint(0, K, (B,), generator=g)
    sub = torch.randint(0, K, (B, T, 1), generator=g)
    for b in range(n_ign):                       # samples b < n_ign carry an ignored past frame
        sub[b, (b * 7) % T, 0] = -1
    lam, ls = 0.3, 0.4
    mix = MixUp(alpha=0.1, label_smoothing={"action": ls}, num_classes={"action": K})

    class _Fixed:
        def sample(self_inner):
            return torch.tensor(lam)
    mix.mixup_beta_sampler = _Fixed()
    x_out, t_out, s_out, ign = mix({m: v.to(dev()) for m, v in feats.items()}, {"action": tgt.to(dev())}, {"action": sub.to(dev())})
    rx, rt_, rs, rign = O.mixup({m: v.clone() for m, v in feats.items()}, tgt, sub, K, ls, lam)
    for m in feats:
        assert rel_l2(x_out[m].cpu(), rx[m]) < 1e-6, m
    assert rel_l2(t_out["action"].cpu(), rt_) < 1e-6
    assert rel_l2(s_out["action"].cpu(), rs) < 1e-6
    assert torch.equal(ign["action"].cpu(), rign)
    if B - n_ign <= 1:                           # no mixing: inputs come back unchanged, labels are plain one-hot
        for m in feats:
            assert torch.equal(x_out[m].cpu(), feats[m])


def test_zero_mask_frames_prologue():
    """ZeroMaskRULSTMFeats on the device: exactly round(T * rate) zero frames per clip, others untouched, subsets differ
    between clips and between calls, and every frame gets masked about equally often."""
    from afft_amd import dropout as D_
    from afft_amd.common.transforms import ZeroMaskRULSTMFeats
    D_.manual_seed(9)
    B, T, C = 64, 16, 40
    x = torch.randn(B, T, C, 1, 1, 1).abs() + 1.0
    tr = ZeroMaskRULSTMFeats(mask_rate=0.2)
    k = round(T * 0.2)
    counts = torch.zeros(T)
    prev = None
    for _ in range(8):
        y = tr(x.clone().to(dev())).cpu()
        zero = (y.reshape(B, T, C) == 0).all(-1)
        assert torch.equal(zero.sum(1), torch.full((B,), k))
        assert torch.equal(y.reshape(B, T, C)[~zero], x.reshape(B, T, C)[~zero])
        assert len({tuple(r.tolist()) for r in zero}) > B // 4          # clips draw different subsets
        if prev is not None:
            assert not torch.equal(prev, zero)                            # a new key per call
        prev = zero
        counts += zero.float().sum(0)
    assert float(counts.min()) > 0.4 * float(counts.mean()) and float(counts.max()) < 1.8 * float(counts.mean())
    assert ZeroMaskRULSTMFeats(0)(x) is x


def test_marginalize_verb_noun_scores():
    from afft_amd.challenge import marginalize_scores
    N, A, V, Nn = 37, 3806, 97, 300
    g = torch.Generator().manual_seed(5)
    logits = torch.randn(N, A, generator=g) * 3
    mv = (torch.rand(A, V, generator=g) < 0.02).float()
    mn = (torch.rand(A, Nn, generator=g) < 0.01).float()
    verb, noun, act = marginalize_scores(logits.to(dev()), {("verb", "action"): mv, ("noun", "action"): mn})
    p = logits.double().softmax(-1)
    assert rel_l2(verb.cpu(), (p @ mv.double()).float()) < 2e-5
    assert rel_l2(noun.cpu(), (p @ mn.double()).float()) < 2e-5
    assert torch.equal(act.cpu(), logits)


def test_errors_are_reported():
    from afft_amd import ops
    a = torch.zeros(4, 4, device=dev())
    with pytest.raises(RuntimeError, match="sequence length"):
        ops.attention_fwd(a, a, a, 1, 129, 1, 4, 1.0, 0, a, None)
    with pytest.raises(RuntimeError, match="period"):
        ops.attention_fwd(a, a, a, 1, 4, 1, 4, 1.0, 3, a, None, mask_period=3)
    with pytest.raises(ValueError):
        ops.gemm(a, torch.zeros(5, 4, device=dev()), a)


# ----------------------------------------------------------------------------- race net of the LDS-ring GEMM kernels
def _lds_pressure_mix(stream, bufs, rounds=1):
    """What shares a CU with a 128-KiB GEMM workgroup inside the training step: the SMALL-register kernels of the other stream
    (<= 16 VGPRs fit beside two 248-VGPR waves per SIMD: LayerNorm-backward's column reduction, the column-sum combine, ordered sums,
    sum of squares) and the LDS users that run on the CUs a partial round leaves free (cast with its LDS transpose, LayerNorm
    backward).  Round 4's LEAD = 7 schedule of the ping-pong kernel lost a write-after-read race against exactly this mix
    (profiles/r04_pp_war_race.txt); GEMMs + copies on the second stream never exposed it."""
    from afft_amd import ops
    with torch.cuda.stream(stream):
        for _ in range(rounds):
            ops.layernorm_bwd(bufs["dy"], bufs["x"], bufs["w"], bufs["mean"], bufs["rstd"], bufs["dx"], dw=bufs["dw"], db=bufs["db"],
                              accumulate=False, dx_bf16=bufs["dxb"], dcol=bufs["dcol"], dcol_accumulate=False)
            ops.colsum(bufs["dxb"], bufs["dcol"])
            ops.cast(bufs["x"], bufs["xc"], bufs["xt"])
            ops.sumsq(bufs["dx"].view(-1), bufs["ss"])
            ops.colsum(bufs["x"], bufs["dcol"])


RACE_CASES = [
    # (name, variant, layout, M, N, K, packed): every kernel that refills an LDS ring while other waves may still read it
    ("pp_nn_dgrad_proj", 3, "nn", 5120, 2048, 2048, False),      # the launch round 4's race was found in (160 tiles: 96 CUs free for the mix)
    ("pp_nn_full_round", 3, "nn", 8192, 2048, 2048, False),      # 256 tiles: every small-register kernel of the mix lands BESIDE a GEMM workgroup
    ("pp_nt_fwd", 3, "nt", 8192, 2048, 2048, False),
    ("pp_tn_wgrad", 3, "tn", 2048, 8192, 1024, False),
    # whole-tile shapes with an even number of K-tiles run the round-6 steady-state kernel (gemm_bf16_pp2_kernel: the four cases above and
    # the two below); edge tiles / odd K-tile counts stay on the general kernel (gemm_bf16_pp_kernel): both are screened
    ("pp2_tn_wgrad_long_k", 3, "tn", 2048, 6144, 5120, False),
    ("pp2_nt_fwd_qkv", 3, "nt", 5120, 6144, 2048, False),
    ("pp_nt_edge_tiles_general", 3, "nt", 5056, 2048, 2048, False),
    ("pp_nn_odd_ktiles_general", 3, "nn", 5120, 2048, 1984, False),
    ("g128_nt", 1, "nt", 1024, 2048, 2048, False),
    # whole 128x128 tiles with an even K-tile count per slice run the round-6 steady-state kernel (gemm_bf16_g2_kernel); its first build let a
    # wave pass the epilogue barrier with fragment reads in flight (NaN losses in half of the bench processes): unsplit layouts, long and short K
    ("g2_nt_unsplit", 1, "nt", 1024, 8192, 2048, False),
    ("g2_nn_unsplit", 1, "nn", 1024, 6144, 2048, False),
    ("g2_tn_short_k", 1, "tn", 2048, 2048, 256, False),
    ("g128_nt_edge_general", 1, "nt", 1088, 3840, 2048, False),
    ("g128_nn_splitk", 1, "nn", 1024, 2048, 2048, False),
    ("bd_nt_packed", 0, "nt", 5120, 2048, 8192, True),
    # the fp16x2 forward instantiations of the ping-pong kernel: "lo8" = fp16 hi segment + block-scaled fp8 lo segment (two K loops over one
    # ring: X3 = 3), "f16x2" = two fp16 segments (X3 = 2)
    ("pp_nt_fp16_lo8", 3, "nt", 5120, 2048, 2048, "lo8"),                 # whole tiles, K-tile count % 4 == 0: the steady-state kernel (pp2, X3 = 3)
    ("pp2_nt_fp16_lo8_long_k", 3, "nt", 5120, 2048, 8192, "lo8"),
    ("pp_nt_fp16_lo8_edge_general", 3, "nt", 5056, 2048, 2048, "lo8"),  # edge tiles: the general kernel's two loops
    ("pp_nt_fp16x2", 3, "nt", 5120, 2048, 2048, "f16x2"),
]


@pytest.mark.parametrize("case", RACE_CASES, ids=lambda c: c[0])
def test_lds_ring_kernels_are_bitwise_stable_under_lds_pressure(case):
    """2000 launches of each LDS-ring GEMM kernel beside the LDS-pressure mix on a second stream, every launch
    compared bitwise with launch 0 (on the device: one counter, no host sync in the loop).  tools/race_net.sh builds the round-3
    schedule (-DAFFT_PP_LEAD=7 -DAFFT_PP_ALLOW_RACY_LEAD) and runs this test against it through AFFT_LIB."""
    from afft_amd import _lib, ops
    name, variant, layout, M, N, K, packed = case
    a_t, b_t = layout[0] == "t", layout[1] == "t"
    A = bfr(rnd(K, M, seed=71) if a_t else rnd(M, K, seed=71)).to(torch.bfloat16).to(dev())
    Bm = bfr(rnd(N, K, seed=72) if b_t else rnd(K, N, seed=72)).to(torch.bfloat16).to(dev())
    pk = None
    planes = packed if isinstance(packed, str) else None
    if planes:
        packed = False
    elif packed:
        pk = torch.empty(N * K, dtype=torch.bfloat16, device=dev())
        ops.pack_weight(Bm.float(), pk)
    odt = torch.float32 if a_t else torch.bfloat16
    outs = [torch.empty(M, N, dtype=odt, device=dev()) for _ in range(4)]
    rows, d = 2048, 2048
    bufs = dict(dy=rnd(rows, d, seed=1).to(torch.bfloat16).to(dev()), x=rnd(rows, d, seed=2).to(dev()), w=rnd(d, seed=3).to(dev()),
                mean=torch.zeros(rows, device=dev()), rstd=torch.ones(rows, device=dev()), dx=torch.empty(rows, d, device=dev()),
                dw=torch.empty(d, device=dev()), db=torch.empty(d, device=dev()), dxb=torch.empty(rows, d, dtype=torch.bfloat16, device=dev()),
                dcol=torch.empty(d, device=dev()), xc=torch.empty(rows, d, dtype=torch.bfloat16, device=dev()),
                xt=torch.empty(d, rows, dtype=torch.bfloat16, device=dev()), ss=torch.zeros(1, device=dev()))
    side = torch.cuda.Stream()
    bad = torch.zeros((), dtype=torch.int64, device=dev())
    _lib.check(_lib.lib().afft_set_gemm_variant(variant))
    try:
        def launch(o):
            ops.gemm(A, Bm, o, a_t=a_t, b_t=b_t, b_packed=pk)
        if planes:      # operands of the fp16x2 forward: fp16 hi plane (+ e4m3 / fp16 lo plane) of the activation, FP16 (+ e4m3) weight image
            a32, w32 = A.float(), Bm.float()
            hi = torch.empty(M, K, dtype=torch.float16, device=dev())
            a8 = torch.empty(M, K, dtype=torch.uint8, device=dev())
            ops.quant_e4m3(a32, 2048.0, a8, hi=hi)
            w16 = w32.half()
            w8 = torch.empty(N, K, dtype=torch.uint8, device=dev())
            ops.quant_e4m3(w32, 256.0, w8)
            sp = ops.Split(a32, f16=True)
            if planes == "lo8":
                assert _lib.lib().afft_gemm_lo8_ok(M, N, K) == 1

                def launch(o):
                    ops.gemm(hi, w16, o, b_t=True, a8=a8, b8=w8)
            else:
                def launch(o):
                    ops.gemm(sp, w16, o, b_t=True)
            odt = torch.float32
            outs = [torch.empty(M, N, dtype=odt, device=dev()) for _ in range(4)]
        first = torch.empty(M, N, dtype=odt, device=dev())
        launch(first)
        torch.cuda.synchronize()
        ref = ((A.float().t() if a_t else A.float()).double() @ (Bm.float().t() if b_t else Bm.float()).double()).float()
        assert rel_l2(first.float().cpu(), ref.cpu()) < 5e-3
        n = 2000
        for i in range(0, n, 4):
            _lds_pressure_mix(side, bufs)
            for o in outs:
                launch(o)
            for o in outs:
                bad += (o != first).sum()
        torch.cuda.synchronize()
        assert int(bad) == 0, f"{name}: {int(bad)} elements differed from launch 0 over {n} launches"
    finally:
        torch.cuda.synchronize()
        _lib.check(_lib.lib().afft_set_gemm_variant(0))


# ----------------------------------------------------------------------------- round 5: glue kernels that replaced torch-native launches
@pytest.mark.parametrize("clips,L,n0,C", [(64, 17, 16, 3806), (3, 5, 1, 11), (5, 7, 7, 40)])
def test_softmax_ce_on_frame_slices_lands_both_gradients_in_one_buffer(clips, L, n0, C):
    """afft_softmax_ce_frames: the two halves x[:, :n0], x[:, n0:] of a (clips, L, C) classifier output (leading dimension padded to 64)
    are walked where they lie, and their gradients are written into the halves of one (clips, L, C) buffer -- against torch on the
    flattened copies (what the reference's MultiDimCrossEntropy does, common/runner.py:13-37)."""
    from afft_amd import ops
    ld = ((C + 63) // 64) * 64
    buf = torch.zeros(clips, L, ld)
    buf[:, :, :C] = rnd(clips, L, C, seed=71, scale=3.0)
    x = buf.to(dev())[:, :, :C]
    dx = torch.full((clips, L, C), 9.0, device=dev())
    g = torch.Generator().manual_seed(5)
    for lo, hi in ((0, n0), (n0, L)):
        n = hi - lo
        if n == 0:
            continue
        labels = torch.randint(0, C, (clips * n,), generator=g)
        labels[::3] = -1
        row_g = rnd(clips * n, seed=72).abs() + 0.1
        rl = torch.empty(clips * n, device=dev())
        ops.softmax_ce_frames(x[:, lo:hi], C, labels=labels.to(dev()), row_loss=rl)
        ops.softmax_ce_frames(x[:, lo:hi], C, labels=labels.to(dev()), row_g=row_g.to(dev()), dlogits3=dx[:, lo:hi])
        ref_in = buf[:, lo:hi, :C].reshape(-1, C).double().requires_grad_(True)
        ref = torch.nn.functional.cross_entropy(ref_in, labels, ignore_index=-1, reduction="none")
        (ref * row_g.double()).sum().backward()
        torch.cuda.synchronize()
        assert rel_l2(rl.cpu(), ref.float()) < 1e-5
        assert rel_l2(dx[:, lo:hi].cpu().reshape(-1, C), ref_in.grad.float()) < 1e-5
    assert not bool((dx == 9.0).any())          # every element of the shared buffer was written


@pytest.mark.parametrize("B,Ta,Tb,C,a_lo,b_lo,nt", [(64, 16, 16, 2048, 1, 1, 15), (3, 4, 6, 64, 1, 2, 3), (2, 5, 5, 8, 0, 0, 5)])
def test_mse_between_frame_ranges_writes_whole_gradients(B, Ta, Tb, C, a_lo, b_lo, nt):
    """afft_mse_loss overwrites its scalar (no zero fill in front) and afft_mse_frames_bwd writes the FULL gradients of both tensors,
    zeros outside the compared frames (common/runner.py:164-166 compares [:, 1:])"""
    from afft_amd import ops
    a, b = rnd(B, Ta, C, seed=81), rnd(B, Tb, C, seed=82)
    ad, bd = a.to(dev()), b.to(dev())
    av = torch.as_strided(ad, (B, nt * C), (ad.stride(0), 1), a_lo * C)
    bv = torch.as_strided(bd, (B, nt * C), (bd.stride(0), 1), b_lo * C)
    loss = torch.full((), 123.0, device=dev())
    ops.mse_loss(av, bv, 1.0 / (B * nt * C), loss)
    ar, br = a.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = ((ar[:, a_lo:a_lo + nt] - br[:, b_lo:b_lo + nt]) ** 2).mean()
    (ref * 0.7).backward()
    da, db = torch.full_like(ad, 5.0), torch.full_like(bd, 5.0)
    ops.mse_frames_bwd(ad, bd, a_lo, b_lo, nt, 1.0 / (B * nt * C), torch.tensor(0.7, device=dev()), da, db)
    torch.cuda.synchronize()
    assert abs(float(loss) - float(ref)) < 2e-6 * abs(float(ref))
    assert rel_l2(da.cpu(), ar.grad.float()) < 1e-6 and rel_l2(db.cpu(), br.grad.float()) < 1e-6
    assert float(da.cpu()[:, :a_lo].abs().sum()) == 0.0 and float(db.cpu()[:, :b_lo].abs().sum()) == 0.0
    only_a = torch.full_like(ad, 5.0)
    ops.mse_frames_bwd(ad, bd, a_lo, b_lo, nt, 1.0 / (B * nt * C), torch.tensor(0.7, device=dev()), only_a, None)
    torch.cuda.synchronize()
    assert torch.equal(only_a, da)


def test_gather_frames_is_cat_slice_backward_and_token0():
    """afft_gather_frames against torch: concatenation along frames, the summed backward of overlapping slices, token 0 of every frame and
    its zero-filled backward; sources with strides of their own (views)"""
    from afft_amd import ops
    B, T, C, k = 5, 6, 72, 2
    z, zh = rnd(B, T, C, seed=91).to(dev()), rnd(B, T - 1 + k, C, seed=92).to(dev())
    whole = torch.full((B, T + k, C), 3.0, device=dev())
    ops.gather_frames(whole, [(z, 0, 1, 0), (zh, 1, T + k, -1)])
    assert torch.equal(whole, torch.cat([z[:, :1], zh], 1))
    gw, gp, gf = rnd(B, T + k, C, seed=93).to(dev()), rnd(B, T, C, seed=94).to(dev()), rnd(B, k, C, seed=95).to(dev())
    tot = gw.clone(); tot[:, :T] += gp; tot[:, T:] += gf
    dzh = torch.full_like(zh, 3.0)
    ops.gather_frames(dzh, [(gw, 0, T - 1 + k, 1), (gp, 0, T - 1, 1), (gf, T - 1, T - 1 + k, 1 - T)])
    assert rel_l2(dzh.cpu(), tot[:, 1:].cpu()) < 1e-6
    dz = torch.full_like(z, 3.0)
    ops.gather_frames(dz, [(gw, 0, 1, 0), (gp, 0, 1, 0)])
    ref = torch.zeros_like(z); ref[:, :1] = tot[:, :1]
    assert rel_l2(dz.cpu(), ref.cpu()) < 1e-6 and float(dz[:, 1:].abs().sum()) == 0.0
    # token 0 of every frame of a [rows * S, d] stream, from a strided view, and back
    S, d = 5, 128
    wide = rnd(B * T * S, d + 64, seed=96).to(dev())
    X = wide[:, :d]                                         # row stride d + 64
    y = torch.empty(B * T, d, device=dev())
    ops.gather_frames(y.view(B * T, 1, d), [(torch.as_strided(X, (B * T, S, d), (S * X.stride(0), X.stride(0), 1)), 0, 1, 0)])
    assert torch.equal(y, X[::S])
    dX = torch.full((B * T * S, d), 3.0, device=dev())
    ops.gather_frames(dX.view(B * T, S, d), [(y.view(B * T, 1, d), 0, 1, 0)])
    ref = torch.zeros_like(dX); ref[::S] = y
    assert torch.equal(dX, ref)
    ops.gather_frames(dX.view(B * T, S, d), [])              # no source: zeros
    assert float(dX.abs().sum()) == 0.0

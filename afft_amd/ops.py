"""Tensor-level wrappers over the C-ABI (include/afft_hip.h).  PyTorch is used for device memory and
streams only; every computation below runs in the hand-written HIP kernels of libafft_hip.so."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import _lib as L
from ._lib import (ACT_DGELU_ERF, ACT_DGELU_TANH, ACT_GELU_ERF, ACT_GELU_TANH, ACT_NONE, BF16, F16, F32, MASK_CAUSAL,
                   MASK_DIAG, MASK_NONE)

Tensor = torch.Tensor
_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}


def _dt(t: Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"afft_amd: unsupported dtype {t.dtype}")


def _p(t: Optional[Tensor]):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("afft_amd: tensors must live on the GPU (there is no CPU path)")
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """raw hipStream_t of torch's current stream (the C-level getter: torch.cuda.current_stream() costs ~10 us of Python per
    launch, which matters once the host is the bottleneck -- the M = 1024 configurations)"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _rowmajor(t: Tensor, what: str):
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise ValueError(f"afft_amd: {what} must be a 2-D tensor with unit column stride, got {tuple(t.shape)} {t.stride()}")
    return t.stride(0)


_WS: dict = {}      # (device index, raw stream) -> split-K scratch of that stream (afft_gemm_t.workspace), never shared
_WS_BYTES = (68 << 20)    # header + every split-K problem of the 128x128 kernel the automatic mode picks (<= 256 tiles x 4 slices x 64 KiB = 64 MiB).
                          # A launch whose problem needs more than this runs unsplit (afft_gemm_workspace_bytes tells how much it wants).


def set_workspace_bytes(n: int):
    """Size of the per-stream split-K scratch.  A launch whose problem needs more than it is given simply runs unsplit."""
    global _WS_BYTES
    if n != _WS_BYTES:
        _WS_BYTES = int(n)
        if _WS and torch.cuda.is_available():
            torch.cuda.synchronize()      # kernels on other streams may still be using the scratch that is about to be freed
        _WS.clear()


def gemm_workspace(device, raw_stream: Optional[int] = None) -> Tensor:
    """The split-K scratch of the CURRENT stream (or of `raw_stream`) on `device`: allocated once per stream (counters
    zeroed once; every launch leaves them zero), owned here so that the library itself never allocates (include/afft_hip.h)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           _stream() if raw_stream is None else raw_stream)
    ws = _WS.get(key)
    if ws is None:
        ws = torch.zeros(_WS_BYTES, dtype=torch.uint8, device=device)
        _WS[key] = ws
    return ws


class Split(object):
    """fp32 matrix [rows, cols] as two 16-bit planes hi + lo, each [pad64(rows), pad64(cols)] with zero tails: the operand of a
    bf16x3 GEMM (bf16 planes, afft_split_bf16) in either role (k-contiguous rows or k-strided columns), or -- f16 = True -- of
    the fp16 two-pass forward GEMM (fp16 planes, afft_split_f16; afft_gemm_t.split3 = 2)."""
    __slots__ = ("planes", "rows", "cols", "f16")

    def __init__(self, x: Tensor, f16: bool = False):
        assert x.dtype == torch.float32 and x.dim() == 2
        rows, cols = x.shape
        pr, pc = (rows + 63) // 64 * 64, (cols + 63) // 64 * 64
        self.rows, self.cols, self.f16 = rows, cols, bool(f16)
        self.planes = torch.empty(2, pr, pc, dtype=torch.float16 if f16 else torch.bfloat16, device=x.device)
        fn = L.lib().afft_split_f16 if f16 else L.lib().afft_split_bf16
        L.check(fn(_p(x), _rowmajor(x, "x"), rows, cols, _p(self.planes), pc, pr, pr * pc, _stream()), "split")


def gemm(a, b, out: Tensor, *, a_t: bool = False, b_t: bool = False, bias: Optional[Tensor] = None,
         act: int = ACT_NONE, aux: Optional[Tensor] = None, pre: Optional[Tensor] = None,
         rowscale: Optional[Tensor] = None, residual: Optional[Tensor] = None, accumulate: bool = False,
         out2: Optional[Tensor] = None, alpha: float = 1.0, drop: Optional["L.Dropout"] = None,
         sgd: Optional["L.SgdFused"] = None, b_packed: Optional[Tensor] = None,
         out_lo: int = 0, a8: Optional[Tensor] = None, b8: Optional[Tensor] = None, out_lo8: Optional[Tensor] = None) -> Tensor:
    """out = epilogue(alpha * A @ B).  A = a (or a.T if a_t), B = b (or b.T if b_t); a, b 2-D views -- or both `Split`
    (two-plane bf16 splits of fp32 matrices): the bf16x3 GEMM, fp32-accurate products on the bf16 MFMA path; or a an fp16 `Split`
    and b a plain fp16 matrix (a weight's FP16 image): the fp16 two-pass forward GEMM (afft_gemm_t.split3 = 2); a and b both plain fp16
    matrices: one fp16 pass (split3 = 4: the activation's hi plane alone, runtime.one_pass_sites).
    b_packed: the fragment-packed copy of a weight `b` [N, K] used as b.T (pack_weight; afft_gemm_t.b_packed).
    out_lo: `out` (fp16) is the hi plane of a two-plane split of the result, the lo plane sits out_lo elements behind it.
    a8 / b8 (uint8, the shapes of a / b): the lo pass on the block-scaled fp8 MFMA (afft_gemm_t.split3 = 3): a = the fp16 HI plane of the
    activation (a plain fp16 matrix), a8 = e4m3(2^11 (x - hi)), b = the weight's FP16 image, b8 = e4m3(2^8 w); out_lo8: the result's lo
    part as an e4m3 byte plane (pitch = out's element pitch)."""
    d = L.GemmDesc()
    if a8 is not None:
        if not (a.dtype == b.dtype == torch.float16 and a8.dtype == b8.dtype == torch.uint8 and b_t and not a_t):
            raise TypeError("afft_amd.gemm: the fp8 lo pass takes fp16 a / b (NT) with uint8 e4m3 planes a8 / b8")
        assert a8.shape == a.shape and b8.shape == b.shape and a8.stride(1) == 1 and b8.stride(1) == 1
        M, K = a.shape
        Kb, N = b.shape[1], b.shape[0]
        d.split3 = 3
        d.a8, d.a8_ld, d.b8, d.b8_ld = _p(a8), a8.stride(0), _p(b8), b8.stride(0)
    elif isinstance(a, Split) and not isinstance(b, Split):
        if not (a.f16 and isinstance(b, torch.Tensor) and b.dtype == torch.float16) or a_t:
            raise TypeError("afft_amd.gemm: a Split A with a plain B is the fp16 two-pass forward GEMM (fp16 planes, fp16 B, A not transposed)")
        sa = a
        a = sa.planes[0]
        M, K = sa.rows, a.shape[1]
        Kb, N = (b.shape[1], b.shape[0]) if b_t else (b.shape[0], b.shape[1])
        d.split3, d.a_lo, d.b_lo = 2, a.numel(), 0
    elif isinstance(a, Split):
        sa, sb = a, b
        a, b = sa.planes[0], sb.planes[0]
        M, K = (sa.cols, a.shape[0]) if a_t else (sa.rows, a.shape[1])
        Kb, N = (b.shape[1], sb.rows) if b_t else (b.shape[0], sb.cols)
        if sa.f16 != sb.f16:
            raise TypeError("afft_amd.gemm: one operand is split into fp16 planes, the other into bf16 planes")
        d.split3, d.a_lo, d.b_lo = (2 if sa.f16 else 1), a.numel(), b.numel()
    else:
        M, K = (a.shape[1], a.shape[0]) if a_t else (a.shape[0], a.shape[1])
        Kb, N = (b.shape[1], b.shape[0]) if b_t else (b.shape[0], b.shape[1])
        if a.dtype == torch.float16 and b.dtype == torch.float16:      # ONE fp16 pass (afft_gemm_t.split3 = 4): forward layouts, like the two-pass GEMM
            d.split3 = 4
    if K != Kb:
        raise ValueError(f"afft_amd.gemm: inner sizes differ ({K} vs {Kb})")
    if a.dtype != b.dtype:
        raise TypeError("afft_amd.gemm: operand dtypes differ")
    if out.shape[0] != M or out.shape[1] != N:
        raise ValueError(f"afft_amd.gemm: out is {tuple(out.shape)}, expected ({M},{N})")
    d.M, d.N, d.K, d.dtype = M, N, K, (L.BF16 if d.split3 else _dt(a))      # split planes: 16-bit elements either way
    d.A = _p(a)
    d.a_rs, d.a_cs = (a.stride(1), a.stride(0)) if a_t else (a.stride(0), a.stride(1))
    d.B = _p(b)
    d.b_rs, d.b_cs = (b.stride(1), b.stride(0)) if b_t else (b.stride(0), b.stride(1))
    if a.dtype in (torch.bfloat16, torch.float16) and d.split3 != 1:      # split-K scratch: plain bf16 and the fp16 two-pass GEMMs
        ws = gemm_workspace(a.device)
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
    if b_packed is not None:
        assert b_packed.dtype == torch.bfloat16 and b_packed.is_contiguous() and b_t and not a_t and b_packed.numel() == N * K
        d.b_packed = _p(b_packed)
    if sgd is not None:       # afft_sgd_fused_t: the result is a weight gradient consumed by the update of sgd.p (layout of `out`)
        d.sgd = C.pointer(sgd)
    d.alpha = alpha
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == N and bias.is_contiguous()
    d.bias = _p(bias)
    d.act = act
    if aux is not None:
        d.aux, d.ldaux, d.aux_dtype = _p(aux), _rowmajor(aux, "aux"), _dt(aux)
    if pre is not None:
        d.pre, d.ldpre, d.pre_dtype = _p(pre), _rowmajor(pre, "pre"), _dt(pre)
    if rowscale is not None:
        assert rowscale.dtype == torch.float32 and rowscale.numel() == M
    d.rowscale = _p(rowscale)
    if residual is not None:
        assert residual.dtype == torch.float32
        d.residual, d.ldres = _p(residual), _rowmajor(residual, "residual")
    d.accumulate = 1 if accumulate else 0
    d.out, d.ldo, d.out_dtype = _p(out), _rowmajor(out, "out"), _dt(out)
    if out_lo:
        assert out.dtype == torch.float16 and out_lo % 8 == 0
        d.out_lo = int(out_lo)
    if out_lo8 is not None:
        assert out.dtype == torch.float16 and out_lo8.dtype == torch.uint8 and out_lo8.stride(0) == out.stride(0) and out_lo8.shape == out.shape
        d.out_lo8 = _p(out_lo8)
    if out2 is not None:
        d.out2, d.ldo2, d.out2_dtype = _p(out2), _rowmajor(out2, "out2"), _dt(out2)
    if drop is not None:
        d.drop = drop
    L.check(L.lib().afft_gemm(C.byref(d), _stream()), "gemm")
    return out


def layernorm_fwd(x: Tensor, w: Optional[Tensor], b: Optional[Tensor], eps: float, y: Tensor,
                  mean: Optional[Tensor] = None, rstd: Optional[Tensor] = None) -> Tensor:
    """x fp32 [rows,d] view (row stride free), y [rows,d] fp32/bf16."""
    assert x.dtype == torch.float32
    rows, d = x.shape
    L.check(L.lib().afft_layernorm_fwd(_p(x), _rowmajor(x, "x"), _p(w), _p(b), eps, rows, d, _p(y),
                                       _rowmajor(y, "y"), _dt(y), _p(mean), _p(rstd), _stream()), "layernorm_fwd")
    return y


def quant_e4m3(src: Tensor, scale: float, dst: Tensor, hi: Optional[Tensor] = None) -> Tensor:
    """dst (uint8 [rows_pad, ldd]) = e4m3(scale * src), zero-filled outside src's shape; with hi (fp16 [rows_pad, ldd]): hi = fp16(src) and
    dst = e4m3(scale * (src - hi)) -- include/afft_hip.h afft_quant_e4m3"""
    assert src.dtype == torch.float32 and dst.dtype == torch.uint8 and dst.dim() == 2 and dst.stride(1) == 1 and dst.is_contiguous()
    if hi is not None:
        assert hi.dtype == torch.float16 and hi.shape == dst.shape and hi.is_contiguous()
    rows, cols = src.shape
    L.check(L.lib().afft_quant_e4m3(_p(src), _rowmajor(src, "src"), rows, cols, float(scale), _p(dst), dst.shape[1], dst.shape[0], _p(hi),
                                    _stream()), "quant_e4m3")
    return dst


def layernorm_fwd_split(x: Tensor, w: Optional[Tensor], b: Optional[Tensor], eps: float, y_hi: Tensor, y_lo: int,
                        y_bf16: Optional[Tensor] = None, mean: Optional[Tensor] = None, rstd: Optional[Tensor] = None,
                        y_lo8: Optional[Tensor] = None) -> Tensor:
    """LayerNorm whose result is written as two fp16 planes (hi at y_hi, lo y_lo elements behind it) -- or hi + an e4m3 byte plane y_lo8 --
    and, optionally, as a bf16 copy."""
    assert x.dtype == torch.float32 and y_hi.dtype == torch.float16 and (y_bf16 is None or y_bf16.dtype == torch.bfloat16)
    rows, d = x.shape
    L.check(L.lib().afft_layernorm_fwd_split(_p(x), _rowmajor(x, "x"), _p(w), _p(b), eps, rows, d, _p(y_hi), _rowmajor(y_hi, "y_hi"),
                                             int(y_lo), _p(y_bf16), _rowmajor(y_bf16, "y_bf16") if y_bf16 is not None else 0,
                                             _p(mean), _p(rstd), _p(y_lo8), _stream()), "layernorm_fwd_split")
    return y_hi


def layernorm_bwd(dy: Tensor, x: Tensor, w: Optional[Tensor], mean: Tensor, rstd: Tensor, dx_out: Tensor,
                  dx_in: Optional[Tensor] = None, dx_bf16: Optional[Tensor] = None, dw: Optional[Tensor] = None,
                  db: Optional[Tensor] = None, accumulate: bool = True, copy_drop: Optional["L.Dropout"] = None,
                  dcol: Optional[Tensor] = None, dcol_accumulate: bool = True) -> Tensor:
    """dw / db (optional) receive the weight / bias gradients: added to when `accumulate`, else overwritten.
    dx_bf16 (optional): bf16 copy of dx_out with the mask `copy_drop` replayed; dcol (optional): its column sums."""
    rows, d = x.shape
    nparts = L.lib().afft_layernorm_bwd_nparts(rows)
    partial = torch.empty(nparts * 3 * d, dtype=torch.float32, device=x.device)
    lddx = _rowmajor(dx_out, "dx_out")
    if dx_in is not None:
        assert _rowmajor(dx_in, "dx_in") == lddx
    if dx_bf16 is not None:
        assert dx_bf16.dtype == torch.bfloat16 and _rowmajor(dx_bf16, "dx_bf16") == lddx
    L.check(L.lib().afft_layernorm_bwd(_p(dy), _rowmajor(dy, "dy"), _dt(dy), _p(x), _rowmajor(x, "x"), _p(w),
                                       _p(mean), _p(rstd), rows, d, _p(dx_in), _p(dx_out), lddx, _p(dx_bf16),
                                       C.byref(copy_drop) if copy_drop is not None else None,
                                       _p(dw), _p(db), 1 if accumulate else 0, _p(dcol), 1 if dcol_accumulate else 0,
                                       _p(partial), _stream()), "layernorm_bwd")
    return dx_out


def attention_fwd(q: Tensor, k: Tensor, v: Tensor, nseq: int, L_: int, H: int, hd: int, scale: float, mask: int,
                  out: Tensor, probs: Optional[Tensor], drop_p: float = 0.0, drop_key: int = 0,
                  mask_period: int = 0) -> Tensor:
    assert q.dtype == k.dtype == v.dtype == out.dtype
    L.check(L.lib().afft_attention_fwd(_p(q), _rowmajor(q, "q"), _p(k), _rowmajor(k, "k"), _p(v), _rowmajor(v, "v"),
                                       _dt(q), nseq, L_, H, hd, scale, mask, mask_period, drop_p, drop_key, _p(out),
                                       _rowmajor(out, "out"), _p(probs), _stream()), "attention_fwd")
    return out


def attention_fwd_table(q: Tensor, k: Tensor, v: Tensor, nseq: int, L_: int, H: int, hd: int, scale: float, table: Tensor,
                        out: Tensor, probs: Optional[Tensor], drop_p: float = 0.0, drop_key: int = 0) -> Tensor:
    """attention with an arbitrary additive fp32 [L, L] mask (afft_attention_fwd_table)"""
    assert q.dtype == k.dtype == v.dtype == out.dtype
    assert table.dtype == torch.float32 and table.shape == (L_, L_) and table.is_contiguous() and table.device == q.device
    L.check(L.lib().afft_attention_fwd_table(_p(q), _rowmajor(q, "q"), _p(k), _rowmajor(k, "k"), _p(v), _rowmajor(v, "v"),
                                             _dt(q), nseq, L_, H, hd, scale, _p(table), drop_p, drop_key, _p(out),
                                             _rowmajor(out, "out"), _p(probs), _stream()), "attention_fwd_table")
    return out


def attention_fwd_split(q: Tensor, k: Tensor, v: Tensor, in_lo: int, nseq: int, L_: int, H: int, hd: int, scale: float, mask: int,
                        out_hi: Tensor, out_lo: int, out_bf16: Optional[Tensor], probs: Optional[Tensor], drop_p: float = 0.0,
                        drop_key: int = 0, mask_period: int = 0, out_lo8: Optional[Tensor] = None) -> Tensor:
    """fp16x2 forward attention: q / k / v / out_hi are the hi planes of two-plane fp16 splits (lo planes in_lo / out_lo elements behind)"""
    assert q.dtype == k.dtype == v.dtype == out_hi.dtype == torch.float16
    L.check(L.lib().afft_attention_fwd_split(_p(q), _rowmajor(q, "q"), _p(k), _rowmajor(k, "k"), _p(v), _rowmajor(v, "v"), int(in_lo),
                                             nseq, L_, H, hd, scale, mask, mask_period, drop_p, drop_key, _p(out_hi),
                                             _rowmajor(out_hi, "out"), int(out_lo), _p(out_bf16),
                                             _rowmajor(out_bf16, "out_bf16") if out_bf16 is not None else 0, _p(probs), _p(out_lo8), _stream()),
            "attention_fwd_split")
    return out_hi


def attention_bwd(dout: Tensor, q: Tensor, k: Tensor, v: Tensor, probs: Tensor, nseq: int, L_: int, H: int, hd: int,
                  scale: float, dq: Tensor, dk: Tensor, dv: Tensor, drop_p: float = 0.0, drop_key: int = 0):
    assert dout.dtype == q.dtype == k.dtype == v.dtype == dq.dtype == dk.dtype == dv.dtype
    L.check(L.lib().afft_attention_bwd(_p(dout), _rowmajor(dout, "dout"), _p(q), _rowmajor(q, "q"), _p(k),
                                       _rowmajor(k, "k"), _p(v), _rowmajor(v, "v"), _dt(q), _p(probs), nseq, L_, H,
                                       hd, scale, drop_p, drop_key, _p(dq), _rowmajor(dq, "dq"), _p(dk),
                                       _rowmajor(dk, "dk"), _p(dv),
                                       _rowmajor(dv, "dv"), _stream()), "attention_bwd")


def softmax_ce(logits: Tensor, C_: int, *, labels: Optional[Tensor] = None, soft: Optional[Tensor] = None,
               keep: Optional[Tensor] = None, gscale: float = 1.0, row_g: Optional[Tensor] = None,
               loss_sum: Optional[Tensor] = None, dlogits: Optional[Tensor] = None,
               row_loss: Optional[Tensor] = None):
    """logits fp32 [rows, >=C] view. labels int64 [rows] (-1 ignored) or soft fp32 [rows, C]; keep uint8 [rows]."""
    rows = logits.shape[0]
    assert logits.dtype == torch.float32
    if labels is not None:
        assert labels.dtype == torch.int64 and labels.numel() == rows and labels.is_contiguous()
    if soft is not None:
        assert soft.dtype == torch.float32 and soft.shape[0] == rows
    if keep is not None:
        assert keep.dtype == torch.uint8 and keep.numel() == rows
    L.check(L.lib().afft_softmax_ce(_p(logits), _rowmajor(logits, "logits"), rows, C_, _p(labels), _p(soft),
                                    _rowmajor(soft, "soft") if soft is not None else 0, _p(keep), gscale,
                                    _p(row_g), _p(loss_sum), _p(dlogits),
                                    _rowmajor(dlogits, "dlogits") if dlogits is not None else 0,
                                    _dt(dlogits) if dlogits is not None else 0, _p(row_loss), _stream()), "softmax_ce")


def softmax_ce_frames(logits3: Tensor, C_: int, *, labels: Optional[Tensor] = None, soft: Optional[Tensor] = None,
                      keep: Optional[Tensor] = None, row_g: Optional[Tensor] = None, dlogits3: Optional[Tensor] = None,
                      row_loss: Optional[Tensor] = None):
    """softmax_ce on a (clips, frames, >= C) VIEW (e.g. x[:, :n] of a wider tensor): no flattening copy; dlogits3 is a view of the
    same kind (its own strides).  Row r of labels / soft / keep / row_g / row_loss = frame r % frames of clip r // frames."""
    clips, frames = logits3.shape[0], logits3.shape[1]
    assert logits3.dtype == torch.float32 and logits3.dim() == 3 and logits3.stride(2) == 1
    rows = clips * frames
    if labels is not None:
        assert labels.dtype == torch.int64 and labels.numel() == rows and labels.is_contiguous()
    if soft is not None:
        assert soft.dtype == torch.float32 and soft.shape[0] == rows
    if keep is not None:
        assert keep.dtype == torch.uint8 and keep.numel() == rows
    if dlogits3 is not None:
        assert dlogits3.dim() == 3 and dlogits3.shape[:2] == logits3.shape[:2] and dlogits3.stride(2) == 1
    L.check(L.lib().afft_softmax_ce_frames(_p(logits3), logits3.stride(0), logits3.stride(1), clips, frames, C_, _p(labels), _p(soft),
                                           _rowmajor(soft, "soft") if soft is not None else 0, _p(keep), 1.0, _p(row_g),
                                           _p(dlogits3), dlogits3.stride(0) if dlogits3 is not None else 0,
                                           dlogits3.stride(1) if dlogits3 is not None else 0,
                                           _dt(dlogits3) if dlogits3 is not None else 0, _p(row_loss), _stream()), "softmax_ce_frames")


def loss_reduce(vals, weights, means: Optional[Tensor], total: Tensor):
    """means[i] = mean(vals[i]), total[0] = sum_i weights[i] * means[i]: Runner._reduce_loss in one launch (afft_loss_reduce)"""
    n = len(vals)
    assert 1 <= n <= 8 and all(v.dtype == torch.float32 and v.is_contiguous() for v in vals)
    ptrs = (C.c_void_p * n)(*[v.data_ptr() for v in vals])
    cnts = (C.c_int64 * n)(*[v.numel() for v in vals])
    ws = (C.c_float * n)(*[float(w) for w in weights])
    L.check(L.lib().afft_loss_reduce(ptrs, cnts, ws, n, _p(means), _p(total), _stream()), "loss_reduce")


def loss_reduce_bwd(grads, weights, g_total: Optional[Tensor], total: Optional[Tensor] = None, ok: Optional[Tensor] = None):
    """grads[i][:] = g_total * weights[i] / grads[i].numel() (None entries are skipped); ok (device float, optional) =
    isfinite(total) and isfinite(g_total): what the optimizer kernels of the step look at (afft_loss_reduce_bwd_ok)"""
    n = len(grads)
    ptrs = (C.c_void_p * n)(*[None if g is None else g.data_ptr() for g in grads])
    cnts = (C.c_int64 * n)(*[0 if g is None else g.numel() for g in grads])
    ws = (C.c_float * n)(*[float(w) for w in weights])
    L.check(L.lib().afft_loss_reduce_bwd_ok(ptrs, cnts, ws, n, _p(g_total), _p(total), _p(ok), _stream()), "loss_reduce_bwd")


def mse(a: Tensor, b: Tensor, gscale: float, loss_sum: Optional[Tensor], da: Optional[Tensor], db: Optional[Tensor],
        g_dev: Optional[Tensor] = None, lscale: float = 1.0):
    """loss_sum[0] += lscale * sum (a - b)^2 (ordered sum through the current stream's scratch: no float atomics), da / db +=
    the gradient."""
    rows, d = a.shape
    assert a.dtype == b.dtype == torch.float32 and b.shape == a.shape
    ws = gemm_workspace(a.device) if loss_sum is not None else None
    L.check(L.lib().afft_mse(_p(a), _rowmajor(a, "a"), _p(b), _rowmajor(b, "b"), rows, d, gscale, _p(g_dev), lscale,
                             _p(loss_sum),
                             _p(da), _rowmajor(da, "da") if da is not None else 0, _p(db),
                             _rowmajor(db, "db") if db is not None else 0, _p(ws), ws.numel() if ws is not None else 0,
                             _stream()), "mse")


def mse_loss(a: Tensor, b: Tensor, lscale: float, loss: Tensor):
    """loss[0] = lscale * sum (a - b)^2, written (ordered sum through the current stream's scratch)"""
    rows, d = a.shape
    assert a.dtype == b.dtype == torch.float32 and b.shape == a.shape
    ws = gemm_workspace(a.device)
    L.check(L.lib().afft_mse_loss(_p(a), _rowmajor(a, "a"), _p(b), _rowmajor(b, "b"), rows, d, lscale, _p(loss), _p(ws), ws.numel(),
                                  _stream()), "mse_loss")


def mse_frames_bwd(a: Tensor, b: Tensor, a_lo: int, b_lo: int, nt: int, gscale: float, g_dev: Optional[Tensor],
                   da: Optional[Tensor], db: Optional[Tensor]):
    """a (B, Ta, C), b (B, Tb, C) fp32 with contiguous frames; da / db contiguous tensors of the same shapes, written whole:
    the gradient of mean-free sum((a[:, a_lo:a_lo+nt] - b[:, b_lo:b_lo+nt])^2) * gscale * g_dev inside the ranges, zeros outside."""
    B, Ta, C_ = a.shape
    Tb = b.shape[1]
    assert a.dtype == b.dtype == torch.float32 and a.stride(2) == 1 and b.stride(2) == 1 and a.stride(1) == C_ and b.stride(1) == C_
    assert (da is None or (da.is_contiguous() and da.shape == a.shape)) and (db is None or (db.is_contiguous() and db.shape == b.shape))
    L.check(L.lib().afft_mse_frames_bwd(_p(a), a.stride(0), a_lo * C_, Ta * C_, _p(b), b.stride(0), b_lo * C_, Tb * C_, B, nt * C_,
                                        gscale, _p(g_dev), _p(da), _p(db), _stream()), "mse_frames_bwd")


def cast(src: Tensor, dst: Optional[Tensor], dst_t: Optional[Tensor] = None, zero_pad: bool = False,
         drop: Optional["L.Dropout"] = None):
    """fp32 [rows, cols] -> dst [rows, >=cols] and/or dst_t [cols, >=rows] (dtype of dst)."""
    rows, cols = src.shape
    assert src.dtype == torch.float32
    ref = dst if dst is not None else dst_t
    L.check(L.lib().afft_cast(_p(src), _rowmajor(src, "src"), rows, cols, _p(dst),
                              dst.stride(0) if dst is not None else 0, _dt(ref), _p(dst_t),
                              dst_t.stride(0) if dst_t is not None else 0, 1 if zero_pad else 0,
                              C.byref(drop) if drop is not None else None, _stream()), "cast")


def assemble_tokens(feats: Sequence[Tensor], token: Tensor, tok_stride_t: int, mod_embed: Optional[Tensor], BT: int,
                    T: int, d: int, X: Tensor):
    n = len(feats)
    ptrs = (C.c_void_p * n)(*[_p(f) for f in feats])
    lds = (C.c_int64 * n)(*[_rowmajor(f, "feat") for f in feats])
    for f in feats:
        assert f.dtype == torch.float32 and f.shape == (BT, d)
    assert X.is_contiguous() and X.dtype == torch.float32
    L.check(L.lib().afft_assemble_tokens(ptrs, lds, n, _p(token), tok_stride_t, _p(mod_embed), BT, T, d, _p(X),
                                         _stream()), "assemble_tokens")
    return X


def colsum(src: Tensor, out: Tensor, accumulate: bool = False):
    """out[n] (+)= sum_m src[m, n] without float atomics (bit-identical from run to run); the row blocks' partial sums go through
    the current stream's scratch (gemm_workspace)."""
    rows, cols = src.shape
    assert out.dtype == torch.float32 and out.numel() >= cols
    ws = gemm_workspace(src.device)
    L.check(L.lib().afft_colsum(_p(src), _rowmajor(src, "src"), _dt(src), rows, cols, _p(out),
                                1 if accumulate else 0, _p(ws), ws.numel(), _stream()), "colsum")
    return out


def gather_frames(out: Tensor, srcs):
    """out (clips, frames, C) fp32 (any clip / frame strides, contiguous C) = sum over srcs of their frames, zeros where none covers:
    srcs = [(tensor (clips, f_k, C), lo, hi, off), ...]: source k contributes src_k[:, t + off] to out[:, t] for lo <= t < hi."""
    clips, frames, C_ = out.shape
    n = len(srcs)
    assert n <= 4 and out.dtype == torch.float32 and out.stride(2) == 1
    for t, lo, hi, off in srcs:
        assert t.dtype == torch.float32 and t.dim() == 3 and t.shape[0] == clips and t.shape[2] == C_ and t.stride(2) == 1
        assert 0 <= lo <= hi <= frames and lo + off >= 0 and (hi == lo or hi - 1 + off < t.shape[1])
    ptrs = (C.c_void_p * max(n, 1))(*[t.data_ptr() for t, *_ in srcs])
    cs = (C.c_int64 * max(n, 1))(*[t.stride(0) for t, *_ in srcs])
    fs = (C.c_int64 * max(n, 1))(*[t.stride(1) for t, *_ in srcs])
    lo = (C.c_int32 * max(n, 1))(*[s[1] for s in srcs])
    hi = (C.c_int32 * max(n, 1))(*[s[2] for s in srcs])
    off = (C.c_int32 * max(n, 1))(*[s[3] for s in srcs])
    L.check(L.lib().afft_gather_frames(_p(out), out.stride(0), out.stride(1), clips, frames, C_, n, ptrs, cs, fs, lo, hi, off, _stream()),
            "gather_frames")
    return out


def add_rows_periodic(x: Tensor, table: Tensor, period: int, y: Tensor):
    rows, d = x.shape
    L.check(L.lib().afft_add_rows_periodic(_p(x), _rowmajor(x, "x"), _p(table), _rowmajor(table, "table"), rows,
                                           period, d, _p(y), _rowmajor(y, "y"), _stream()), "add_rows_periodic")
    return y


def reduce_rows_periodic(src: Tensor, period: int, out: Tensor):
    rows, d = src.shape
    L.check(L.lib().afft_reduce_rows_periodic(_p(src), _rowmajor(src, "src"), rows, period, d, _p(out),
                                              _rowmajor(out, "out"), _stream()), "reduce_rows_periodic")
    return out


def sgd_nesterov(p: Tensor, g: Tensor, buf: Tensor, lr: float, mom: float, wd: float, gscale: float, first,
                 p_bf16: Optional[Tensor] = None, gscale_dev: Optional[Tensor] = None, p_f16: Optional[Tensor] = None,
                 p_f8: Optional[Tensor] = None, ok: Optional[Tensor] = None):
    """ok (device float): the update is applied iff ok[0] != 0 (loss_reduce writes isfinite(loss) there).
    first: bool (first step) or the AFFT_SGD_* flag word (1 = first step, 2 = plain momentum instead of Nesterov);
    p_bf16 / p_f16: the bf16 / fp16 images of the updated weights (same element offsets as p)"""
    assert p.is_contiguous() and g.is_contiguous() and buf.is_contiguous()
    if p_bf16 is not None:
        assert p_bf16.dtype == torch.bfloat16 and p_bf16.numel() == p.numel() and p_bf16.is_contiguous()
    if p_f16 is not None:
        assert p_f16.dtype == torch.float16 and p_f16.numel() == p.numel() and p_f16.is_contiguous()
    if p_f8 is not None:
        assert p_f8.dtype == torch.uint8 and p_f8.numel() == p.numel() and p_f8.is_contiguous()
    L.check(L.lib().afft_sgd_nesterov2(_p(p), _p(g), _dt(g), _p(buf), _p(p_bf16), _p(p_f16), _p(p_f8), p.numel(), lr, mom, wd, gscale,
                                       _p(gscale_dev), int(first), _p(ok), _stream()), "sgd_nesterov")


def sgd_nesterov_runs(p: Tensor, g: Tensor, buf: Tensor, runs: Tensor, lr: float, mom: float, wd: float, gscale: float, first: bool,
                      p_bf16: Optional[Tensor] = None, p_f16: Optional[Tensor] = None, p_f8: Optional[Tensor] = None,
                      ok: Optional[Tensor] = None):
    """the same update over the runs {start, length} (int64 [nruns, 2] on the device) of whole flat buffers"""
    assert runs.dtype == torch.int64 and runs.dim() == 2 and runs.shape[1] == 2 and runs.is_contiguous()
    assert p.dtype == g.dtype == buf.dtype == torch.float32
    L.check(L.lib().afft_sgd_nesterov_runs2(_p(p), _p(g), _p(buf), _p(p_bf16), _p(p_f16), _p(p_f8), _p(runs), runs.shape[0], lr, mom, wd, gscale,
                                            int(first), _p(ok), _stream()), "sgd_nesterov_runs")


def pack_weight(w: Tensor, dst: Tensor) -> Tensor:
    """fp32 weight [rows, cols] (row stride free; rows % 16 == 0, cols % 32 == 0) -> its fragment-packed bf16 image `dst`
    (rows * cols elements, contiguous): include/afft_hip.h afft_pack_weight"""
    assert w.dtype == torch.float32 and w.dim() == 2 and w.stride(1) == 1
    assert dst.dtype == torch.bfloat16 and dst.is_contiguous() and dst.numel() == w.numel()
    L.check(L.lib().afft_pack_weight(_p(w), w.stride(0), w.shape[0], w.shape[1], _p(dst), _stream()), "pack_weight")
    return dst


def sumsq(x: Tensor, out: Tensor, scale: float = 1.0):
    """out[0] += scale * sum(x^2) over a flat fp32 / bf16 buffer (ordered sum through the current stream's scratch)."""
    assert x.is_contiguous() and out.dtype == torch.float32
    ws = gemm_workspace(x.device)
    L.check(L.lib().afft_sumsq(_p(x), _dt(x), x.numel(), scale, _p(out), _p(ws), ws.numel(), _stream()), "sumsq")
    return out


def clip_coef(sumsq_: Tensor, max_norm: float, coef: Tensor, norm_out: Optional[Tensor] = None):
    """coef[0] = min(1, max_norm / (sqrt(sumsq[0]) + 1e-6)) on the device (no host sync)."""
    L.check(L.lib().afft_clip_coef(_p(sumsq_), float(max_norm), _p(coef), _p(norm_out), _stream()), "clip_coef")
    return coef


def group_sum(x: Tensor, G: int, S: int, W: int, scale: float, y: Tensor):
    """y[g, :] = scale * sum_s x[g, s, :] for contiguous fp32 x [G, S, W]."""
    assert x.is_contiguous() and y.is_contiguous() and x.dtype == y.dtype == torch.float32
    L.check(L.lib().afft_group_sum(_p(x), G, S, W, scale, _p(y), _stream()), "group_sum")
    return y


def group_bcast(dy: Tensor, G: int, S: int, W: int, scale: float, dx: Tensor):
    """dx[g, s, :] = scale * dy[g, :]."""
    assert dy.is_contiguous() and dx.is_contiguous() and dy.dtype == dx.dtype == torch.float32
    L.check(L.lib().afft_group_bcast(_p(dy), G, S, W, scale, _p(dx), _stream()), "group_bcast")
    return dx


def act_bwd(act: int, dy: Tensor, saved: Optional[Tensor], dpre: Tensor, aux: Optional[Tensor] = None,
            daux: Optional[Tensor] = None, drop: Optional["L.Dropout"] = None):
    """dpre = d/dpre [drop(act(pre))] * dy (see afft_act_bwd)."""
    rows, cols = dy.shape
    assert dy.dtype == torch.float32
    L.check(L.lib().afft_act_bwd(act, _p(dy), _rowmajor(dy, "dy"), _p(saved), _rowmajor(saved, "saved") if saved is not None else 0,
                                 _dt(saved) if saved is not None else F32, _p(aux),
                                 _rowmajor(aux, "aux") if aux is not None else 0, rows, cols,
                                 C.byref(drop) if drop is not None else None, _p(dpre), _rowmajor(dpre, "dpre"), _dt(dpre),
                                 _p(daux), _rowmajor(daux, "daux") if daux is not None else 0, _stream()), "act_bwd")
    return dpre


def softmax_small_fwd(x: Tensor, y: Tensor):
    rows, n = x.shape
    L.check(L.lib().afft_softmax_small_fwd(_p(x), _rowmajor(x, "x"), rows, n, _p(y), _rowmajor(y, "y"), _stream()), "softmax_small_fwd")
    return y


def softmax_small_bwd(y: Tensor, dy: Tensor, dx: Tensor):
    rows, n = y.shape
    L.check(L.lib().afft_softmax_small_bwd(_p(y), _rowmajor(y, "y"), _p(dy), _rowmajor(dy, "dy"), rows, n, _p(dx),
                                           _rowmajor(dx, "dx"), _stream()), "softmax_small_bwd")
    return dx


def _ptr_array(ts):
    arr = (C.c_void_p * len(ts))()
    for i, t in enumerate(ts):
        arr[i] = t.data_ptr() if t is not None else None
    return arr


def weighted_sum_fwd(xs: Sequence[Tensor], w: Tensor, out: Tensor):
    """out[r, :] = sum_i w[r, i] * xs[i][r, :]; all xs share one row stride."""
    rows, cols = xs[0].shape
    ld = _rowmajor(xs[0], "x")
    assert all(_rowmajor(x, "x") == ld and x.shape == xs[0].shape and x.dtype == torch.float32 for x in xs)
    L.check(L.lib().afft_weighted_sum_fwd(_ptr_array(xs), ld, _p(w), _rowmajor(w, "w"), len(xs), rows, cols, _p(out),
                                          _rowmajor(out, "out"), _stream()), "weighted_sum_fwd")
    return out


def weighted_sum_bwd(xs: Sequence[Tensor], w: Tensor, dout: Tensor, dxs: Sequence[Tensor], dw: Tensor):
    rows, cols = xs[0].shape
    ld = _rowmajor(xs[0], "x")
    ldd = _rowmajor(dxs[0], "dx")
    assert all(_rowmajor(x, "dx") == ldd for x in dxs)
    L.check(L.lib().afft_weighted_sum_bwd(_ptr_array(xs), ld, _p(w), _rowmajor(w, "w"), _p(dout), _rowmajor(dout, "dout"),
                                          len(xs), rows, cols, _ptr_array(dxs), ldd, _p(dw), _rowmajor(dw, "dw"),
                                          _stream()), "weighted_sum_bwd")


def mixup_plan(labels_subclips: Optional[Tensor], B: int, ignore_cls: int, partner: Tensor, ignore_mask: Optional[Tensor] = None):
    T = 0
    if labels_subclips is not None:
        assert labels_subclips.dtype == torch.int64 and labels_subclips.is_contiguous() and labels_subclips.shape[0] == B
        T = labels_subclips.numel() // B
    assert partner.dtype == torch.int32 and partner.numel() == B
    L.check(L.lib().afft_mixup_plan(_p(labels_subclips), B, T, ignore_cls, _p(partner), _p(ignore_mask), _stream()), "mixup_plan")
    return partner


def mixup_rows(x: Tensor, partner: Tensor, lam: float, y: Tensor):
    B = x.shape[0]
    assert x.is_contiguous() and y.is_contiguous() and x.dtype == y.dtype == torch.float32
    L.check(L.lib().afft_mixup_rows(_p(x), B, x.numel() // B, _p(partner), lam, _p(y), _stream()), "mixup_rows")
    return y


def mixup_labels(labels: Tensor, B: int, K: int, label_smooth: float, ignore_cls: int, partner: Tensor, lam: float, out: Tensor):
    assert labels.dtype == torch.int64 and labels.is_contiguous() and out.is_contiguous() and out.dtype == torch.float32
    L.check(L.lib().afft_mixup_labels(_p(labels), B, labels.numel() // B, K, label_smooth, ignore_cls, _p(partner), lam,
                                      _p(out), _stream()), "mixup_labels")
    return out


def softmax_rows(x: Tensor, y: Tensor):
    rows, Cc = x.shape
    assert x.dtype == y.dtype == torch.float32
    L.check(L.lib().afft_softmax_rows(_p(x), _rowmajor(x, "x"), rows, Cc, _p(y), _rowmajor(y, "y"), _stream()), "softmax_rows")
    return y


def zero_mask_frames(x: Tensor, k: int, key: int):
    """x fp32 [B, T, ...] contiguous: zero k random frames of every clip in place."""
    assert x.is_contiguous() and x.dtype == torch.float32 and x.dim() >= 2
    B, T = x.shape[0], x.shape[1]
    L.check(L.lib().afft_zero_mask_frames(_p(x), B, T, x.numel() // max(B * T, 1), k, key & 0xFFFFFFFF, _stream()), "zero_mask_frames")
    return x

"""TEST DOUBLE of afft_amd.ops -- test infrastructure, never part of the product.

The product has no CPU path (afft_amd.ops raises for non-GPU tensors).  To exercise the HOST logic of the package in the
build container -- the autograd wiring of afft_amd.functional (gradient sink, gradient hand-over, readiness notifications),
afft_amd.parallel (flat buffers, bucket protocol, fused-SGD call pattern, world_size-2 gloo all-reduce) and the mirrored
reference modules -- `installed()` swaps every tensor-level wrapper of afft_amd.ops for a plain-torch restatement of the
C-ABI contract written in include/afft_hip.h, for the duration of a `with` block.  Dropout descriptors are rejected (the CPU
tests run with every rate = 0); kernels themselves are only ever tested on the GPU (tests/test_kernels_gpu.py).
"""
from __future__ import annotations

import contextlib
import math

import torch

from afft_amd import _lib as L

F = torch.nn.functional


def _no_drop(d):
    if d is not None and (d.p > 0 or d.path_p > 0):
        raise NotImplementedError("cpu_ops test double: dropout is not modelled")


def _act(act, v, aux):
    if act == L.ACT_NONE:
        return v
    if act == L.ACT_GELU_ERF:
        return F.gelu(v)
    if act == L.ACT_GELU_TANH:
        return F.gelu(v, approximate="tanh")
    if act in (L.ACT_DGELU_ERF, L.ACT_DGELU_TANH):
        a = aux.float().detach().requires_grad_(True)
        with torch.enable_grad():
            y = F.gelu(a) if act == L.ACT_DGELU_ERF else F.gelu(a, approximate="tanh")
            (g,) = torch.autograd.grad(y.sum(), a)
        return v * g
    if act == L.ACT_RELU:
        return v.clamp_min(0)
    if act == L.ACT_SIGMOID_GATE:
        return aux.float() * torch.sigmoid(v)
    raise ValueError(act)


class Split:
    """ops.Split: x = hi + lo in two zero-padded 16-bit planes (bf16, or fp16 for the fp16x2 forward mode)"""

    def __init__(self, x, f16=False):
        rows, cols = x.shape
        pr, pc = (rows + 63) // 64 * 64, (cols + 63) // 64 * 64
        self.rows, self.cols, self.f16 = rows, cols, bool(f16)
        dt = torch.float16 if f16 else torch.bfloat16
        self.planes = torch.zeros(2, pr, pc, dtype=dt)
        hi = x.detach().to(dt)
        self.planes[0, :rows, :cols] = hi
        self.planes[1, :rows, :cols] = (x.detach() - hi.float()).to(dt)


@torch.no_grad()
def gemm(a, b, out, *, a_t=False, b_t=False, bias=None, act=L.ACT_NONE, aux=None, pre=None, rowscale=None, residual=None,
         accumulate=False, out2=None, alpha=1.0, drop=None, sgd=None, b_packed=None, out_lo=0):
    assert sgd is None and b_packed is None and not out_lo, "cpu_ops test double: fused update / packed weights / plane outputs are GPU-only paths"
    _no_drop(drop)
    if isinstance(a, Split) and not isinstance(b, Split):     # fp16 two-pass forward: A = hi + lo planes, B the weight's FP16 image
        assert a.f16 and b.dtype == torch.float16 and not a_t
        ah, al = (p.float() for p in a.planes)
        B = (b.t() if b_t else b).float()
        v = alpha * (ah @ B + al @ B)[:out.shape[0], :out.shape[1]]
    elif isinstance(a, Split):     # bf16x3: hi*hi + lo*hi + hi*lo over the padded planes, live part of the result
        ah, al = (p.float().t() if a_t else p.float() for p in a.planes)
        bh, bl = (p.float().t() if b_t else p.float() for p in b.planes)
        if a.f16:     # fp16x2: the first two segments only (A = hi + lo, B rounded once)
            v = alpha * (ah @ bh + al @ bh)[:out.shape[0], :out.shape[1]]
        else:
            v = alpha * (ah @ bh + al @ bh + ah @ bl)[:out.shape[0], :out.shape[1]]
    else:
        A = (a.t() if a_t else a).float()
        B = (b.t() if b_t else b).float()
        assert A.shape[1] == B.shape[0] and tuple(out.shape) == (A.shape[0], B.shape[1])
        v = alpha * (A @ B)
    if bias is not None:
        v = v + bias
    if pre is not None:
        pre.copy_(v)
    v = _act(act, v, aux)
    if rowscale is not None:
        v = v * rowscale[:, None]
    if residual is not None:
        v = v + residual
    if accumulate:
        v = v + out.float()
    out.copy_(v)
    if out2 is not None:
        out2.copy_(v)
    return out


@torch.no_grad()
def layernorm_fwd(x, w, b, eps, y, mean=None, rstd=None):
    mu = x.mean(1)
    var = x.var(1, unbiased=False)
    rs = torch.rsqrt(var + eps)
    v = (x - mu[:, None]) * rs[:, None]
    if w is not None:
        v = v * w
    if b is not None:
        v = v + b
    y.copy_(v)
    if mean is not None and mean.numel():
        mean.copy_(mu)
        rstd.copy_(rs)
    return y


@torch.no_grad()
def layernorm_bwd(dy, x, w, mean, rstd, dx_out, dx_in=None, dx_bf16=None, dw=None, db=None, accumulate=True,
                  copy_drop=None, dcol=None, dcol_accumulate=True):
    _no_drop(copy_drop)
    dyf = dy.float()
    xhat = (x - mean[:, None]) * rstd[:, None]
    g = dyf * w if w is not None else dyf
    dx = rstd[:, None] * (g - g.mean(1, keepdim=True) - xhat * (g * xhat).mean(1, keepdim=True))
    if dx_in is not None:
        dx = dx + dx_in
    dx_out.copy_(dx)
    if dw is not None:
        t = (dyf * xhat).sum(0)
        dw.copy_(dw + t if accumulate else t)
    if db is not None:
        t = dyf.sum(0)
        db.copy_(db + t if accumulate else t)
    if dx_bf16 is not None:
        dx_bf16.copy_(dx)
        if dcol is not None:
            t = dx_bf16.float().sum(0)
            dcol.copy_(dcol + t if dcol_accumulate else t)
    return dx_out


def _mask(kind, Ln, period):
    if kind == L.MASK_NONE:
        return None
    m = torch.zeros(Ln, Ln)
    if kind == L.MASK_DIAG:
        m.fill_diagonal_(float("-inf"))
    elif kind == L.MASK_CAUSAL:
        m = torch.triu(torch.full((Ln, Ln), float("-inf")), diagonal=1)
    elif kind == L.MASK_BLOCKCAUSAL:
        i = torch.arange(Ln)
        m = torch.where((i[None, :] % period) > (i[:, None] % period), float("-inf"), 0.0)
    return m


def _heads(t, nseq, Ln, H, hd):
    return t.float().reshape(nseq, Ln, H, hd).permute(0, 2, 1, 3)


@torch.no_grad()
def attention_fwd(q, k, v, nseq, L_, H, hd, scale, mask, out, probs, drop_p=0.0, drop_key=0, mask_period=0):
    assert drop_p == 0.0
    qh, kh, vh = (_heads(t, nseq, L_, H, hd) for t in (q, k, v))
    s = qh @ kh.transpose(-1, -2) * scale
    m = _mask(mask, L_, mask_period)
    if m is not None:
        s = s + m
    p = torch.softmax(s, -1)
    if probs is not None:
        probs.copy_(p)
    out.copy_((p @ vh).permute(0, 2, 1, 3).reshape(nseq * L_, H * hd))
    return out


@torch.no_grad()
def attention_fwd_table(q, k, v, nseq, L_, H, hd, scale, table, out, probs, drop_p=0.0, drop_key=0):
    assert drop_p == 0.0
    qh, kh, vh = (_heads(t, nseq, L_, H, hd) for t in (q, k, v))
    p = torch.softmax(qh @ kh.transpose(-1, -2) * scale + table, -1)
    if probs is not None:
        probs.copy_(p)
    out.copy_((p @ vh).permute(0, 2, 1, 3).reshape(nseq * L_, H * hd))
    return out


@torch.no_grad()
def attention_bwd(dout, q, k, v, probs, nseq, L_, H, hd, scale, dq, dk, dv, drop_p=0.0, drop_key=0):
    assert drop_p == 0.0
    qh, kh, vh, doh = (_heads(t, nseq, L_, H, hd) for t in (q, k, v, dout))
    p = probs.float()
    dvh = p.transpose(-1, -2) @ doh
    dp = doh @ vh.transpose(-1, -2)
    ds = p * (dp - (dp * p).sum(-1, keepdim=True)) * scale
    back = lambda t: t.permute(0, 2, 1, 3).reshape(nseq * L_, H * hd)   # noqa: E731
    dq.copy_(back(ds @ kh))
    dk.copy_(back(ds.transpose(-1, -2) @ qh))
    dv.copy_(back(dvh))


@torch.no_grad()
def softmax_ce(logits, C_, *, labels=None, soft=None, keep=None, gscale=1.0, row_g=None, loss_sum=None, dlogits=None,
               row_loss=None):
    x = logits[:, :C_].float()
    rows = x.shape[0]
    lp = torch.log_softmax(x, 1)
    if labels is not None:
        if int(labels.max()) >= C_:
            raise RuntimeError("softmax_ce: label out of range")
        kept = labels >= 0
        tgt = torch.zeros(rows, C_)
        tgt[kept, labels[kept]] = 1.0
    else:
        kept = torch.ones(rows, dtype=torch.bool) if keep is None else keep.bool()
        tgt = soft[:, :C_].float() * kept[:, None]
    rl = -(tgt * lp).sum(1) * kept
    if row_loss is not None:
        row_loss.copy_(rl)
    if loss_sum is not None:
        loss_sum += rl.sum()
    if dlogits is not None:
        g = torch.softmax(x, 1) * tgt.sum(1, keepdim=True) - tgt
        g = g * gscale * kept[:, None]
        if row_g is not None:
            g = g * row_g[:, None]
        dlogits.zero_()
        dlogits[:, :C_].copy_(g)


@torch.no_grad()
def softmax_ce_frames(logits3, C_, *, labels=None, soft=None, keep=None, row_g=None, dlogits3=None, row_loss=None):
    clips, frames = logits3.shape[:2]
    d2 = None if dlogits3 is None else torch.zeros(clips * frames, dlogits3.shape[2])
    softmax_ce(logits3.reshape(clips * frames, -1), C_, labels=labels, soft=soft, keep=keep, row_g=row_g, dlogits=d2, row_loss=row_loss)
    if dlogits3 is not None:
        dlogits3.copy_(d2.view(clips, frames, -1))


def mse_loss(a, b, lscale, loss):
    loss.copy_(lscale * ((a - b) ** 2).sum())


def mse_frames_bwd(a, b, a_lo, b_lo, nt, gscale, g_dev, da, db):
    g = gscale * (float(g_dev) if g_dev is not None else 1.0) * 2.0 * (a[:, a_lo:a_lo + nt] - b[:, b_lo:b_lo + nt])
    if da is not None:
        da.zero_()
        da[:, a_lo:a_lo + nt] = g
    if db is not None:
        db.zero_()
        db[:, b_lo:b_lo + nt] = -g


def mse(a, b, gscale, loss_sum, da, db, g_dev=None, lscale=1.0):
    diff = a - b
    if loss_sum is not None:
        loss_sum += lscale * (diff * diff).sum()
    g = gscale * (float(g_dev) if g_dev is not None else 1.0) * 2.0 * diff
    if da is not None:
        da += g
    if db is not None:
        db -= g


@torch.no_grad()
def cast(src, dst, dst_t=None, zero_pad=False, drop=None):
    _no_drop(drop)
    rows, cols = src.shape
    if dst is not None:
        dst[:rows, :cols].copy_(src)
    if dst_t is not None:
        dst_t[:cols, :rows].copy_(src.t())


@torch.no_grad()
def assemble_tokens(feats, token, tok_stride_t, mod_embed, BT, T, d, X):
    if tok_stride_t:
        X[:, 0].copy_(token.reshape(-1, d)[torch.arange(BT) % T])
    else:
        X[:, 0].copy_(token.reshape(-1, d)[0].expand(BT, d))
    for i, f in enumerate(feats):
        X[:, i + 1].copy_(f)
    if mod_embed is not None:
        X += mod_embed[None]
    return X


@torch.no_grad()
def colsum(src, out, accumulate=False):
    t = src.float().sum(0)
    n = t.numel()
    out[:n].copy_(out[:n] + t if accumulate else t)
    return out


@torch.no_grad()
def gather_frames(out, srcs):
    out.zero_()
    for t, lo, hi, off in srcs:
        if hi > lo:
            out[:, lo:hi] += t[:, lo + off:hi + off]
    return out


@torch.no_grad()
def add_rows_periodic(x, table, period, y):
    y.copy_(x + table[torch.arange(x.shape[0]) % period])
    return y


@torch.no_grad()
def reduce_rows_periodic(src, period, out):
    out.index_add_(0, torch.arange(src.shape[0]) % period, src.float())
    return out


@torch.no_grad()
def sgd_nesterov(p, g, buf, lr, mom, wd, gscale, first, p_bf16=None, gscale_dev=None, p_f16=None, p_f8=None, ok=None):
    if ok is not None and float(ok) == 0.0:
        return
    if gscale_dev is not None:
        gscale = gscale * float(gscale_dev)
    flags = int(first)        # AFFT_SGD_* flag word: 1 = first step, 2 = plain momentum (nesterov=False)
    gg = g.float() * gscale + wd * p
    bb = gg if (flags & 1) else mom * buf + gg
    buf.copy_(bb)
    p -= lr * (bb if (flags & 2) else gg + mom * bb)
    if p_bf16 is not None:
        p_bf16.copy_(p)
    if p_f16 is not None:
        p_f16.copy_(p)


@torch.no_grad()
def sgd_nesterov_runs(p, g, buf, runs, lr, mom, wd, gscale, first, p_bf16=None, p_f16=None, p_f8=None, ok=None):
    if ok is not None and float(ok) == 0.0:
        return
    for a, n in runs.tolist():
        sgd_nesterov(p[a:a + n], g[a:a + n], buf[a:a + n], lr, mom, wd, gscale, first,
                     p_bf16=None if p_bf16 is None else p_bf16[a:a + n], p_f16=None if p_f16 is None else p_f16[a:a + n])


@torch.no_grad()
def loss_reduce(vals, weights, means, total):
    tot = torch.zeros((), dtype=torch.float32)
    for i, (v, w) in enumerate(zip(vals, weights)):
        m = v.float().mean() if v.numel() else torch.zeros(())
        if means is not None:
            means[i] = m
        if w != 0:
            tot = tot + w * m
    total.copy_(tot)


@torch.no_grad()
def loss_reduce_bwd(grads, weights, g_total, total=None, ok=None):
    go = 1.0 if g_total is None else float(g_total)
    if ok is not None:
        fin = (total is None or bool(torch.isfinite(total))) and go == go and abs(go) != float("inf")
        ok.fill_(1.0 if fin else 0.0)
    for g, w in zip(grads, weights):
        if g is not None and g.numel():
            g.fill_(go * w / g.numel())


@torch.no_grad()
def sumsq(x, out, scale=1.0):
    out += scale * (x.float() ** 2).sum()
    return out


@torch.no_grad()
def clip_coef(sumsq_, max_norm, coef, norm_out=None):
    n = math.sqrt(float(sumsq_))
    coef.fill_(min(1.0, max_norm / (n + 1e-6)))
    if norm_out is not None:
        norm_out.fill_(n)
    return coef


@torch.no_grad()
def group_sum(x, G, S, W, scale, y):
    y.copy_(scale * x.reshape(G, S, W).sum(1))
    return y


@torch.no_grad()
def group_bcast(dy, G, S, W, scale, dx):
    dx.reshape(G, S, W).copy_(scale * dy.reshape(G, 1, W).expand(G, S, W))
    return dx


@torch.no_grad()
def act_bwd(act, dy, saved, dpre, aux=None, daux=None, drop=None):
    _no_drop(drop)
    if act == L.ACT_NONE:
        dpre.copy_(dy)
    elif act == L.ACT_RELU:
        dpre.copy_(dy * (saved.float() > 0))
    elif act == L.ACT_GELU_ERF:
        dpre.copy_(_act(L.ACT_DGELU_ERF, dy, saved))
    elif act == L.ACT_GELU_TANH:
        dpre.copy_(_act(L.ACT_DGELU_TANH, dy, saved))
    elif act == L.ACT_SIGMOID_GATE:
        s = torch.sigmoid(saved.float())
        dpre.copy_(dy * aux * s * (1 - s))
        daux.copy_(dy * s)
    else:
        raise ValueError(act)
    return dpre


@torch.no_grad()
def softmax_small_fwd(x, y):
    y.copy_(torch.softmax(x, 1))
    return y


@torch.no_grad()
def softmax_small_bwd(y, dy, dx):
    dx.copy_(y * (dy - (dy * y).sum(1, keepdim=True)))
    return dx


@torch.no_grad()
def weighted_sum_fwd(xs, w, out):
    out.copy_(sum(w[:, i:i + 1] * x for i, x in enumerate(xs)))
    return out


@torch.no_grad()
def weighted_sum_bwd(xs, w, dout, dxs, dw):
    for i, (x, dx) in enumerate(zip(xs, dxs)):
        dx.copy_(w[:, i:i + 1] * dout)
        dw[:, i].copy_((dout * x).sum(1))


@torch.no_grad()
def softmax_rows(x, y):
    y.copy_(torch.softmax(x, 1))
    return y


_NAMES = ["Split", "gemm", "layernorm_fwd", "layernorm_bwd", "attention_fwd", "attention_fwd_table", "attention_bwd", "softmax_ce", "softmax_ce_frames", "mse", "mse_loss", "mse_frames_bwd", "cast",
          "assemble_tokens", "colsum", "gather_frames", "add_rows_periodic", "reduce_rows_periodic", "sgd_nesterov", "sgd_nesterov_runs", "loss_reduce", "loss_reduce_bwd", "sumsq", "clip_coef",
          "group_sum", "group_bcast", "act_bwd", "softmax_small_fwd", "softmax_small_bwd", "weighted_sum_fwd",
          "weighted_sum_bwd", "softmax_rows"]


@contextlib.contextmanager
def installed():
    """Swap the wrappers of afft_amd.ops for the torch restatements above (and back)."""
    from afft_amd import ops
    saved = {n: getattr(ops, n) for n in _NAMES}
    g = globals()
    try:
        for n in _NAMES:
            setattr(ops, n, g[n])
        yield
    finally:
        for n, f in saved.items():
            setattr(ops, n, f)

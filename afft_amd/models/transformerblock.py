"""MI355X mirror of the reference's ``models/transformerblock.py``: same classes, constructor arguments,
parameter names and return values; the arithmetic runs in the HIP kernels (afft_amd.functional).

  Attention       <- models/transformerblock.py:7-36     (returns (x, attn) like the reference)
  CrossAttention  <- models/transformerblock.py:39-76
  MLP             <- models/transformerblock.py:79-93    (Sequential indices 0 and 2 hold the two Linears)
  DropPath        <- models/transformerblock.py:96-115
  Block           <- models/transformerblock.py:118-135  (pre-LN; fused LN->QKV->attention->proj+residual)
  DecoderBlock    <- models/transformerblock.py:138-162
"""
from __future__ import annotations

from typing import Optional, Union

import torch
import torch.nn as nn

from .. import functional as F_
from .. import dropout as D_

Tensor = torch.Tensor
MaskArg = Union[None, str, Tensor]


def mask_kind(attn_mask: MaskArg, n: int):
    """The HIP attention kernels apply their mask in-register from a kind: 'none' | 'diag' | 'causal' |
    ('blockcausal', T).  The reference passes additive -inf tensors (models/fusion.py:30-32,170-171,313-317);
    recognise those.  Any other (N, N) tensor -- the reference adds whatever it is given -- becomes ('table', fp32 tensor);
    masks that broadcast over batch or heads ((B, 1, N, N), ...) are not supported."""
    if attn_mask is None:
        return "none"
    if isinstance(attn_mask, tuple):
        if len(attn_mask) == 2 and attn_mask[0] == "table":
            return attn_mask
        if len(attn_mask) != 2 or attn_mask[0] != "blockcausal" or n % int(attn_mask[1]) != 0:
            raise ValueError(f"unknown mask kind {attn_mask!r}")
        return ("blockcausal", int(attn_mask[1]))
    if isinstance(attn_mask, str):
        if attn_mask not in ("none", "diag", "causal"):
            raise ValueError(f"unknown mask kind {attn_mask!r}")
        return attn_mask
    m = attn_mask.detach().to("cpu", torch.float32)
    if m.shape != (n, n):
        raise ValueError(f"attn_mask must be ({n},{n}), got {tuple(m.shape)}")
    if torch.equal(m, torch.zeros(n, n)):
        return "none"
    causal = lambda t: torch.triu(torch.full((t, t), float("-inf")), diagonal=1)   # noqa: E731
    if torch.equal(m, causal(n)):
        return "causal"
    eye = torch.zeros(n, n)
    eye.fill_diagonal_(float("-inf"))
    if torch.equal(m, eye):
        return "diag"
    for t in range(1, n):            # T-SA-Fuser: the causal T x T mask tiled over the modalities
        if n % t == 0 and torch.equal(m, causal(t).repeat(n // t, n // t)):
            return ("blockcausal", t)
    # anything else: `attn = attn + attn_mask` for an arbitrary (N, N) tensor (models/transformerblock.py:26-28, :66-68) -- the table goes to
    # the generic attention kernel (afft_attention_fwd_table) instead of an in-register mask of the MFMA kernels
    return ("table", attn_mask.detach().to(torch.float32).contiguous())


def _flat(x: Tensor):
    B, N, C = x.shape
    x2 = x.reshape(B * N, C)
    if x2.dtype != torch.float32:
        x2 = x2.float()
    return x2.contiguous(), B, N, C


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def drop_cfg(self):
        return D_.cfg(self, attn=self.attn_drop.p, out=self.proj_drop.p)

    def forward(self, x, attn_mask: MaskArg = None):
        x2, B, N, C = _flat(x)
        y, probs = F_.AttnSublayer.apply(x2, None, None, self.qkv.weight, self.qkv.bias, self.proj.weight,
                                         self.proj.bias, N, self.num_heads, mask_kind(attn_mask, N), 0.0, False,
                                         False, self.scale, self.drop_cfg())
        return y.view(B, N, C), probs


class CrossAttention(nn.Module):
    """Cross attention used in transformer decoder"""

    def __init__(self, dim, mem_dim=None, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        mem_dim = mem_dim or dim      # mem_dim != dim and qkv_bias=True are off the AFFT configurations: served call by call (functional.CrossAttnSublayer)
        self.w_q = nn.Linear(dim, dim, bias=qkv_bias)
        self.w_k = nn.Linear(mem_dim, dim, bias=qkv_bias)
        self.w_v = nn.Linear(mem_dim, dim, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def drop_cfg(self):
        return D_.cfg(self, attn=self.attn_drop.p, out=self.proj_drop.p)

    def forward(self, x, mem, attn_mask: MaskArg = None):
        x2, B, N, C = _flat(x)
        m2, _, _, _ = _flat(mem)
        y = F_.CrossAttnSublayer.apply(x2, m2, None, None, None, None, self.w_q.weight, self.w_k.weight,
                                       self.w_v.weight, self.proj.weight, self.proj.bias, N, self.num_heads,
                                       mask_kind(attn_mask, N), 0.0, False, self.scale, self.drop_cfg(),
                                       self.w_q.bias, self.w_k.bias, self.w_v.bias)
        return y.view(B, N, C)


class MLP(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        if act_layer is not nn.GELU:      # every call site of the reference passes nn.GELU (exact erf: the GEMM epilogue's activation)
            raise NotImplementedError("afft_amd: only nn.GELU (exact erf) is on the AFFT path")
        self.mlp = nn.Sequential(
            nn.Linear(in_features, hidden_features),
            act_layer(),
            nn.Linear(hidden_features, out_features),
            nn.Dropout(drop)
        )

    def drop_cfg(self):
        return D_.cfg(self, out=self.mlp[3].p)

    def forward(self, x):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).float().contiguous()
        y = F_.MLPSublayer.apply(x2, None, None, self.mlp[0].weight, self.mlp[0].bias, self.mlp[2].weight,
                                 self.mlp[2].bias, 0.0, "erf", False, False, self.drop_cfg())
        return y.view(*shp[:-1], y.shape[-1])


def drop_path(x, drop_prob: float = 0., training: bool = False):
    """Per-sample stochastic depth (standalone form; inside Block it is fused into the GEMM epilogue)."""
    if drop_prob == 0. or not training:
        return x
    return D_.drop_path_standalone(x, drop_prob)


class DropPath(nn.Module):
    """Drop paths (Stochastic Depth) per sample (when applied in main path of residual blocks)."""

    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        return drop_path(x, self.drop_prob, self.training)


def _dp_rate(mod) -> float:
    return float(mod.drop_prob) if isinstance(mod, DropPath) and mod.drop_prob and mod.training else 0.0


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                              proj_drop=drop)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        mlp_hidden_dim = int(dim * mlp_ratio)
        self.mlp = MLP(in_features=dim, hidden_features=mlp_hidden_dim, act_layer=act_layer, drop=drop)

    def forward_rows(self, x2: Tensor, L: int, mask: str, probs_out: Optional[Tensor] = None):
        """x2: fp32 [nseq*L, dim] rows. Returns (rows, probs [nseq, H, L, L]); probs_out: where to write the attention maps."""
        a = self.attn
        dp = _dp_rate(self.drop_path)
        x2, probs = F_.AttnSublayer.apply(x2, self.norm1.weight, self.norm1.bias, a.qkv.weight, a.qkv.bias,
                                          a.proj.weight, a.proj.bias, L, a.num_heads, mask, self.norm1.eps, False,
                                          True, a.scale, D_.with_path(a.drop_cfg(), dp, L), probs_out)
        m = self.mlp.mlp
        x2 = F_.MLPSublayer.apply(x2, self.norm2.weight, self.norm2.bias, m[0].weight, m[0].bias, m[2].weight,
                                  m[2].bias, self.norm2.eps, "erf", False, True,
                                  D_.with_path(self.mlp.drop_cfg(), dp, L))
        return x2, probs

    def forward_rows_first_token(self, x2: Tensor, L: int, mask: str, probs_out: Optional[Tensor] = None):
        """forward_rows for a caller that only uses token 0 of every sequence afterwards (the SA-Fuser's last block,
        models/fusion.py:362-365): attention over all L tokens, the MLP half on the nseq token-0 rows only.
        Returns (rows [nseq, dim], probs [nseq, H, L, L]) -- the same numbers as forward_rows(...)[0][::L] whenever the MLP's
        element dropout is off (eval mode, p = 0: the goldens and the on / off test).  In training the dropout mask of an element
        is a function of (key, element index), and token 0 of frame r is row r * L in the full-row run but row r here: the two
        runs draw different (equally distributed) masks for the MLP output, so they agree in distribution, not bit for bit; the
        DropPath group index is the same in both."""
        a = self.attn
        dp = _dp_rate(self.drop_path)
        if F_.attn_take_ok(x2, L, a.num_heads):
            # the output projection, its residual / bias / dropout and everything behind them in backward on the token-0 rows only
            # (4/5 of the projection's forward, data-gradient and weight-gradient work at S = 5 never reaches an output)
            x0, probs = F_.AttnSublayer.apply(x2, self.norm1.weight, self.norm1.bias, a.qkv.weight, a.qkv.bias,
                                              a.proj.weight, a.proj.bias, L, a.num_heads, mask, self.norm1.eps, False,
                                              True, a.scale, D_.with_path(a.drop_cfg(), dp, 1), probs_out, L)
        else:
            x2, probs = F_.AttnSublayer.apply(x2, self.norm1.weight, self.norm1.bias, a.qkv.weight, a.qkv.bias,
                                              a.proj.weight, a.proj.bias, L, a.num_heads, mask, self.norm1.eps, False,
                                              True, a.scale, D_.with_path(a.drop_cfg(), dp, L), probs_out)
            x0 = F_.TakeRows.apply(x2, L)
        m = self.mlp.mlp
        x0 = F_.MLPSublayer.apply(x0, self.norm2.weight, self.norm2.bias, m[0].weight, m[0].bias, m[2].weight,
                                  m[2].bias, self.norm2.eps, "erf", False, True,
                                  D_.with_path(self.mlp.drop_cfg(), dp, 1))
        return x0, probs

    def forward(self, x, attn_mask: MaskArg = None):
        x2, B, N, C = _flat(x)
        y, probs = self.forward_rows(x2, N, mask_kind(attn_mask, N))
        return y.view(B, N, C), probs


class DecoderBlock(nn.Module):
    """Transformer decoder block with pre-layernorm"""

    def __init__(self, dim, mem_dim=None, num_heads=4, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0.,
                 attn_drop=0., drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm_self = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                              proj_drop=drop)
        self.cross_attn = CrossAttention(dim, mem_dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale,
                                         attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm_q = norm_layer(dim)
        self.norm_kv = norm_layer(mem_dim or dim)
        self.norm_mlp = norm_layer(dim)
        mlp_hidden_dim = int(dim * mlp_ratio)
        self.mlp = MLP(in_features=dim, hidden_features=mlp_hidden_dim, act_layer=act_layer, drop=drop)

    def forward_rows(self, x2: Tensor, m2: Tensor, L: int, mask: str) -> Tensor:
        a, c = self.attn, self.cross_attn
        dp = _dp_rate(self.drop_path)
        x2, _ = F_.AttnSublayer.apply(x2, self.norm_self.weight, self.norm_self.bias, a.qkv.weight, a.qkv.bias,
                                      a.proj.weight, a.proj.bias, L, a.num_heads, mask, self.norm_self.eps, False,
                                      True, a.scale, D_.with_path(a.drop_cfg(), dp, L))
        x2 = F_.CrossAttnSublayer.apply(x2, m2, self.norm_q.weight, self.norm_q.bias, self.norm_kv.weight,
                                        self.norm_kv.bias, c.w_q.weight, c.w_k.weight, c.w_v.weight, c.proj.weight,
                                        c.proj.bias, L, c.num_heads, mask, self.norm_q.eps, True, c.scale,
                                        D_.with_path(c.drop_cfg(), dp, L), c.w_q.bias, c.w_k.bias, c.w_v.bias)
        m = self.mlp.mlp
        x2 = F_.MLPSublayer.apply(x2, self.norm_mlp.weight, self.norm_mlp.bias, m[0].weight, m[0].bias, m[2].weight,
                                  m[2].bias, self.norm_mlp.eps, "erf", False, True,
                                  D_.with_path(self.mlp.drop_cfg(), dp, L))
        return x2

    def forward(self, x, mem, attn_mask: MaskArg = None):
        x2, B, N, C = _flat(x)
        m2, _, _, _ = _flat(mem)
        return self.forward_rows(x2, m2, N, mask_kind(attn_mask, N)).view(B, N, C)

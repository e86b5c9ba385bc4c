"""Train-mode dropout bookkeeping.  Masks are never stored: every kernel that applies a mask derives it
from (key, element index) (afft_amd/csrc/common.h: drop_keep), and the backward pass replays the same key.

Dropout sites of the path: token / embedding dropout, attention-probability dropout,
projection and MLP output dropout, DropPath per frame (SA-Fuser) or per clip (CA-Fuser), the GPT-2
embd/attn/resid dropouts and the classifier's Dropout(0.2).  Bitwise parity with torch's RNG stream is neither
possible nor required (parity is checked with every rate = 0); the statistics (keep rate, 1/(1-p) scaling,
per-sample DropPath) follow the reference.
"""
from __future__ import annotations

import itertools
from typing import NamedTuple, Optional

from ._lib import Dropout

_seed = 0x1234ABCD
_counter = itertools.count(1)


def manual_seed(seed: int):
    global _seed, _counter
    _seed = seed & 0xFFFFFFFF
    _counter = itertools.count(1)


_salt = None     # device word (torch.int32[1]) every kernel XORs into its keys once enabled


def enable_device_salt(device) -> "torch.Tensor":
    """Move the step-to-step variation of the dropout masks into device memory (afft_set_dropout_salt): needed by a
    captured hipGraph of the training step, whose kernel arguments -- including the host-drawn keys -- are frozen."""
    global _salt
    import torch
    from . import _lib
    if _salt is None or _salt.device != torch.device(device):
        _salt = torch.zeros(1, dtype=torch.int32, device=device)
        _lib.check(_lib.lib().afft_set_dropout_salt(_salt.data_ptr()), "set_dropout_salt")
    return _salt


def disable_device_salt():
    global _salt
    from . import _lib
    _lib.check(_lib.lib().afft_set_dropout_salt(None), "set_dropout_salt")
    _salt = None


def salt_step():
    """advance the device salt (one tiny kernel on the current stream); no-op while the salt is off"""
    if _salt is not None:
        import torch
        from . import _lib
        _lib.check(_lib.lib().afft_dropout_salt_step(_salt.data_ptr(), torch.cuda.current_stream().cuda_stream),
                   "dropout_salt_step")


def next_key() -> int:
    c = next(_counter)
    x = (_seed * 0x9E3779B1 + c * 0x85EBCA77) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x2C1B3C6D) & 0xFFFFFFFF
    x ^= x >> 12
    return x


class DropCfg(NamedTuple):
    """Per-call dropout description handed to a sub-layer Function (None = no dropout)."""
    p_attn: float = 0.0
    k_attn: int = 0
    p_out: float = 0.0
    k_out: int = 0
    p_path: float = 0.0
    k_path: int = 0
    group: int = 1

    def out_desc(self) -> Optional[Dropout]:
        if self.p_out <= 0.0 and self.p_path <= 0.0:
            return None
        return Dropout(self.p_out, self.k_out, self.p_path, self.k_path, self.group)


def cfg(module, attn: float = 0.0, out: float = 0.0) -> Optional[DropCfg]:
    if not module.training or (attn <= 0.0 and out <= 0.0):
        return None
    return DropCfg(p_attn=float(attn), k_attn=next_key(), p_out=float(out), k_out=next_key())


def with_path(c: Optional[DropCfg], p_path: float, group: int, training: Optional[bool] = None) -> Optional[DropCfg]:
    """Add DropPath (rate p_path, one decision per `group` consecutive rows). The reference's DropPath module is
    an nn.Identity when its rate is 0 and is only active in training: a None cfg with p_path>0 still needs the
    module's training flag, which the caller folds in by passing p_path=0 in eval mode."""
    if p_path <= 0.0:
        return c
    base = c or DropCfg()
    return base._replace(p_path=float(p_path), k_path=next_key(), group=int(group))


def elementwise(p: float) -> Optional[Dropout]:
    """Descriptor for a plain nn.Dropout(p) on a tensor (embedding / classifier-input dropout)."""
    if p <= 0.0:
        return None
    return Dropout(float(p), next_key(), 0.0, 0, 1)


def drop_path_standalone(x, p: float):
    """DropPath applied outside a fused sub-layer (the public DropPath module, models/transformerblock.py:96-115): x is
    (N, ...), one decision per dim-0 sample.  Differentiable: the backward pass replays the same per-sample mask."""
    from . import functional as F_
    n = x.shape[0]
    x2 = x.reshape(n, -1).float().contiguous()
    y = F_.ElementDropout.apply(x2, Dropout(0.0, 0, float(p), next_key(), 1))
    return y.view_as(x)

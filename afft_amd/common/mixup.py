"""MI355X mirror of the reference's ``common/mixup.py``: MixUp with an ignore class (some sequences do not have ground
truth for every past frame) as a GPU prologue of the training step.  Same constructor, call signature and return
values as ``common.mixup.MixUp`` (common/mixup.py:93-182); the work is three HIP kernels and nothing returns to the
host: which samples take part (``afft_mixup_plan``), the mixed features (``afft_mixup_rows``) and the mixed, smoothed
one-hot labels (``afft_mixup_labels``).  The reference's ``if batch_wo_ignore_index.sum() <= 1`` early return (a
device-to-host sync) becomes "every sample is its own partner" inside the plan kernel.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Union

import torch

from .. import ops


def batch_wo_ignore_cls(target_subclips: torch.Tensor, ignore_cls=-1):
    target_subclips = target_subclips.squeeze(-1)
    assert target_subclips.ndim == 2, "Target subclips should have dimension of 2."
    return (target_subclips != ignore_cls).all(-1)


def convert_to_one_hot(targets: torch.Tensor, num_class: int, label_smooth: float = 0.0) -> torch.Tensor:
    """common/mixup.py:17-47 through the label kernel (every sample its own partner)."""
    assert 0 <= label_smooth < 1.0, "Label smooth value needs to be between 0 and 1."
    t = targets.squeeze(-1).contiguous()
    n = t.numel()
    partner = torch.arange(n, dtype=torch.int32, device=t.device)
    out = torch.empty(*t.shape, num_class, dtype=torch.float32, device=t.device)
    ops.mixup_labels(t.view(-1), n, num_class, float(label_smooth), -(2 ** 62), partner, 1.0, out.view(n, num_class))
    return out


class MixUp(torch.nn.Module):
    """Mixup: Beyond Empirical Risk Minimization (https://arxiv.org/abs/1710.09412)"""

    def __init__(self, alpha: float = 1.0, label_smoothing: Dict = 0.0, num_classes: Dict = None, one_hot: bool = False,
                 ignore_cls=-1) -> None:
        super().__init__()
        self.mixup_beta_sampler = torch.distributions.beta.Beta(alpha, alpha)   # sampled on the host: no GPU sync
        self.label_smoothing = label_smoothing
        self.num_classes = num_classes
        # kept for the constructor contract: the reference stores the flag (common/mixup.py:116) and its forward never
        # reads it (labels always arrive as class indices and go through convert_to_one_hot, :133-150)
        self.one_hot = one_hot
        self.ignore_cls = ignore_cls

    def forward(self, x_video: Dict, labels: Dict, labels_subclips: Union[Dict, None]) -> Sequence[Union[Dict, None]]:
        first = next(iter(x_video.values()))
        B = first.size(0)
        assert B > 1, "MixUp cannot be applied to a single instance."
        dev = first.device
        lam = float(self.mixup_beta_sampler.sample())
        partner = torch.empty(B, dtype=torch.int32, device=dev)
        ign_index = None
        if labels_subclips is not None:
            cur = next(iter(labels_subclips.values()))
            sub2 = cur.squeeze(-1)
            assert sub2.ndim == 2, "Target subclips should have dimension of 2."
            ign_u8 = torch.empty(sub2.shape, dtype=torch.uint8, device=dev)
            ops.mixup_plan(sub2.contiguous(), B, self.ignore_cls, partner, ign_u8)
            ign_index = {}
            for key, val in labels_subclips.items():
                ign_index[key] = ign_u8.view(val.squeeze(-1).shape).bool().view(val.shape) if val is cur else (val == self.ignore_cls)
        else:
            ops.mixup_plan(None, B, self.ignore_cls, partner)       # batch_wo_ignore_index = [...]: every sample

        def mix_labels(val, key):
            t = val.squeeze(-1).contiguous()
            K = self.num_classes[key]
            out = torch.empty(*t.shape, K, dtype=torch.float32, device=dev)
            ops.mixup_labels(t.view(-1), B, K, float(self.label_smoothing[key]), self.ignore_cls, partner, lam,
                             out.view(-1, K))
            return out

        labels_out = {key: mix_labels(val, key) for key, val in labels.items()}
        if labels_subclips is None:
            return x_video, labels_out, None, None                  # (sic) the reference returns the unmixed inputs here
        x_out = {}
        for modk, x in x_video.items():
            xc = x.contiguous() if x.dtype == torch.float32 else x.float().contiguous()
            y = torch.empty_like(xc)
            ops.mixup_rows(xc, partner, lam, y)
            x_out[modk] = y
        labels_subclips_out = {key: mix_labels(val, key) for key, val in labels_subclips.items()}
        return x_out, labels_out, labels_subclips_out, ign_index

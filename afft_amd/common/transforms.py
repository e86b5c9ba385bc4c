"""MI355X mirror of the reference's ``common/transforms.py`` for batches that are already on the device.

``ZeroMaskRULSTMFeats`` (common/transforms.py:13-26) zeroes ``round(T * mask_rate)`` random frames of a clip; the
reference applies it per clip on the host inside the dataset (train.py:33-41).  Here it is a GPU prologue on the loader
layout ``(B, T, C, 1, 1, 1)`` / ``(B, T, C)``: one kernel per modality, the random subset comes from a counter hash
(afft_amd.dropout key stream), nothing touches the host.  ``PermuteRULSTMFeats`` is a view."""
from __future__ import annotations

from typing import Dict

import torch

from .. import dropout as D_, ops


class PermuteRULSTMFeats:
    def __call__(self, vid):
        return vid.permute(3, 0, 1, 2)


class ZeroMaskRULSTMFeats:
    """Mask random frames with zeros -- batched, on the device, in place."""

    def __init__(self, mask_rate=0.2):
        self.mask_rate = mask_rate

    def __call__(self, vid: torch.Tensor) -> torch.Tensor:
        """vid: fp32 (B, T, ...) contiguous on the GPU; every clip gets its own random subset of frames."""
        if self.mask_rate == 0:
            return vid
        k = round(vid.size(1) * self.mask_rate)
        return ops.zero_mask_frames(vid, k, D_.next_key())

    def apply_dict(self, feats: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """one independent draw per modality, as the reference's per-modality transform lists (train.py:33-41)"""
        return {m: self(v) for m, v in feats.items()}

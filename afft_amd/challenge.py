"""GPU side of the reference's ``challenge.marginalize_verb_noun`` (challenge.py:196-210): action logits -> softmax ->
verb and noun scores by the class-mapping matrices, as one row-softmax kernel and two exact-fp32 MFMA GEMMs on tensors
that are already on the device (the reference does this in numpy on host copies).  The accuracy bookkeeping
(top-k / mean-top-5-recall over the dataset annotations) stays with the caller."""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from . import ops


def marginalize_scores(res_action: torch.Tensor, class_mappings: Dict[Tuple[str, str], torch.Tensor], to_prob: bool = True):
    """res_action fp32 [N, A] action logits (or probabilities if to_prob=False); class_mappings[('verb','action')] fp32
    [A, V], [('noun','action')] fp32 [A, Nn] (datasets/epic_kitchens.py).  Returns [verb [N,V], noun [N,Nn], action]:
    like the reference, the ACTION scores returned are the inputs themselves, not the probabilities."""
    x = res_action.reshape(-1, res_action.shape[-1])
    x = x if x.dtype == torch.float32 else x.float()
    x = x if x.stride(-1) == 1 else x.contiguous()
    probs = x
    if to_prob:
        probs = torch.empty(x.shape[0], x.shape[1], dtype=torch.float32, device=x.device)
        ops.softmax_rows(x, probs)
    out = []
    for key in (("verb", "action"), ("noun", "action")):
        m = class_mappings[key].to(device=x.device, dtype=torch.float32)
        y = torch.empty(x.shape[0], m.shape[1], dtype=torch.float32, device=x.device)
        ops.gemm(probs, m, y)      # fp32 operands -> exact v_mfma_f32_32x32x2_f32 path
        out.append(y)
    return [out[0], out[1], res_action]

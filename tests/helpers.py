"""Shared helpers for the parity tests (CPU and GPU)."""
from __future__ import annotations

import json
import os

import numpy as np
import torch

import closed_form as cf
from cases import CASES, oracle_cfg

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name: str):
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False)
    shapes = {k: tuple(v) for k, v in json.loads(str(z["shapes"])).items()}
    return z, shapes


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_rel(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def case_tensors(name: str):
    """(case dict, state dict, inputs (B,T,C,1,1,1), target, target_subclips) regenerated from closed form."""
    c = CASES[name]
    _, shapes = load_golden(name)
    state = cf.fill_state(shapes)
    data = cf.inputs_for(name, c["modal_dims"], c["B"], c["T"])
    tgt, sub = cf.labels_for(name, c["B"], c["T"], c["num_classes"], c.get("ignore_frac", 0.25))
    return c, state, data, tgt, sub


def flatten_outputs(out: dict) -> dict:
    flat = {}
    for k, v in out.items():
        if k == "attentions":
            flat["attentions/modality_attns"] = v["all-fused"]["modality_attns"]
            continue
        for kk, t in v.items():
            flat[f"{k}/{kk}"] = t
    return flat


def surrogate(out: dict):
    return (out["logits/action"]["all-fused"].pow(2).mean() + out["past_logits/action"]["all-fused"].pow(2).mean()
            + out["past_futures"]["all-fused"].pow(2).mean())

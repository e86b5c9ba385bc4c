"""Shared helpers for the parity tests (CPU and GPU)."""
from __future__ import annotations

import json
import os

import numpy as np
import torch

import closed_form as cf
from cases import CASES, FULL_CASES, FULL_GRAD_SAMPLES, FULL_MAX_WHOLE, FULL_SAMPLES, oracle_cfg

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


# ---- full-size fixtures (compact storage, cases.FULL_CASES)

def strided_sample(t, n: int):
    """At most n elements of the flattened tensor at a fixed stride (the first of every ceil(numel / n))."""
    flat = t.reshape(-1)
    step = max(1, -(-flat.numel() // n))
    return flat[::step]


def compact_entry(t: torch.Tensor, max_whole: int = FULL_MAX_WHOLE, n: int = FULL_SAMPLES) -> dict:
    """What a full-size fixture keeps of a tensor: all of it if small, else norm, sum and a strided sample."""
    t = t.detach().double().cpu()
    if t.numel() <= max_whole:
        return {"whole": t.float().numpy()}
    return {"norm": np.asarray(float(t.norm())), "sum": np.asarray(float(t.sum())), "sample": strided_sample(t, n).float().numpy()}


def compact_error(got: torch.Tensor, z, prefix: str, n: int = FULL_SAMPLES) -> float:
    """Worst relative deviation of `got` from the fixture entry `prefix` (whole tensor, or norm + strided sample)."""
    got = got.detach().double().cpu()
    if prefix + ":whole" in z.files:
        ref = torch.from_numpy(z[prefix + ":whole"]).double()
        assert tuple(got.shape) == tuple(ref.shape), (prefix, got.shape, ref.shape)
        return float((got - ref).norm() / (ref.norm() + 1e-30))
    rn = float(z[prefix + ":norm"])
    ref = torch.from_numpy(z[prefix + ":sample"]).double()
    smp = strided_sample(got, n)
    assert smp.shape == ref.shape, (prefix, smp.shape, ref.shape)
    return max(abs(float(got.norm()) - rn) / (rn + 1e-30), float((smp - ref).norm() / (ref.norm() + 1e-30)))


def full_case_tensors(name: str):
    """(case dict, fixture, state dict, inputs, target, target_subclips) of a full-size fixture, regenerated from closed form."""
    c = FULL_CASES[name]
    z, shapes = load_golden(name)
    state = cf.fill_state(shapes)
    data = cf.inputs_for(name, c["modal_dims"], c["B"], c["T"])
    tgt, sub = cf.labels_for(name, c["B"], c["T"], c["num_classes"], c.get("ignore_frac", 0.25))
    return c, z, state, data, tgt, sub


def full_gradient_errors(named_grads: dict, z) -> dict:
    """{parameter name: worst relative deviation (norm, 256-element strided sample)} against a full-size fixture."""
    names = [str(s) for s in z["gradnames"]]
    norms, samples, offs = z["gradnorm"], z["gradsamples"], z["gradsample_offsets"]
    out = {}
    for i, nm in enumerate(names):
        g = named_grads[nm].detach().double().cpu()
        ref = torch.from_numpy(samples[offs[i]:offs[i + 1]]).double()
        smp = strided_sample(g, FULL_GRAD_SAMPLES)
        assert smp.shape == ref.shape, (nm, smp.shape, ref.shape)
        # tiny gradients (a bias behind a softmax that ignores it) are compared on the scale of the typical parameter gradient
        scale = max(float(norms[i]), 1e-6 * float(np.median(norms)))
        out[nm] = max(abs(float(g.norm()) - float(norms[i])) / scale,
                      float((smp - ref).norm()) / max(float(ref.norm()), scale * (smp.numel() / g.numel()) ** 0.5))
    return out


# ---- interface edges (tests/golden/e0_edges.npz: the reference's Block with an arbitrary additive mask, DecoderBlock(mem_dim != dim, qkv_bias))
def edge_fixture():
    """(npz, meta dict, {'block': state, 'dec': state}, inputs dict) -- weights and inputs regenerated from closed form"""
    z = np.load(os.path.join(GOLDEN, "e0_edges.npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    shapes = json.loads(str(z["shapes"]))
    states = {tag: {k: cf.tensor_for(f"e0.{tag}.{k}", tuple(shp)) for k, shp in shapes[tag].items()} for tag in ("block", "dec")}
    N, L, d, dm = meta["N"], meta["L"], meta["d"], meta["mem_dim"]
    inputs = {"block.x": cf.tensor_for("e0.block.x", (N, L, d), "input"), "dec.x": cf.tensor_for("e0.dec.x", (N, L, d), "input"),
              "dec.mem": cf.tensor_for("e0.dec.mem", (N, L, dm), "input"),
              "block.mask": torch.from_numpy(z["block.mask"]), "dec.mask": torch.from_numpy(z["dec.mask"])}
    return z, meta, states, inputs


def run_edge_modules(device, states, inputs, meta):
    """the mirrored Block / DecoderBlock on `device` -> {name: tensor} with the fixture's keys (outputs, attention, every gradient)"""
    from afft_amd.models.transformerblock import Block, DecoderBlock
    from afft_amd import runtime as rt
    d, dm, H = meta["d"], meta["mem_dim"], meta["heads"]
    out = {}
    rt.SINK.begin_step()
    blk = Block(d, H).eval()
    blk.load_state_dict(states["block"])
    blk = blk.to(device)
    x = inputs["block.x"].to(device).requires_grad_(True)
    y, attn = blk(x, inputs["block.mask"].to(device))
    y.pow(2).mean().backward()
    rt.SINK.finish_step(list(blk.parameters()))
    out.update({"block.y": y, "block.attn": attn, "block.dx": x.grad})
    out.update({f"block.grad.{k}": p.grad for k, p in blk.named_parameters()})
    rt.SINK.begin_step()
    dec = DecoderBlock(d, mem_dim=dm, num_heads=H, qkv_bias=True).eval()
    dec.load_state_dict(states["dec"])
    dec = dec.to(device)
    x2 = inputs["dec.x"].to(device).requires_grad_(True)
    mem = inputs["dec.mem"].to(device).requires_grad_(True)
    y2 = dec(x2, mem, inputs["dec.mask"].to(device))
    y2.pow(2).mean().backward()
    rt.SINK.finish_step(list(dec.parameters()))
    out.update({"dec.y": y2, "dec.dx": x2.grad, "dec.dmem": mem.grad})
    out.update({f"dec.grad.{k}": p.grad for k, p in dec.named_parameters()})
    return out


def edge_error(got: torch.Tensor, ref: torch.Tensor) -> float:
    """relative L2 -- except for a gradient that is analytically ZERO (the k-projection bias of an attention layer: a constant added to
    every key moves no softmax weight), which the fixture holds as 1e-9 rounding noise: there, the absolute size of what we computed"""
    ref = ref.detach().double().cpu()
    got = got.detach().double().cpu()
    if float(ref.abs().max()) < 1e-7:
        return float(got.abs().max())
    return float((got - ref).norm() / ref.norm())

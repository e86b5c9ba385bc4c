"""Closed-form (hash-based, bit-exact on every machine) weights and inputs for golden fixtures.

Both sides of a parity check regenerate the same tensors from (name, shape) alone, so the
fixtures only need to store *outputs*.  Pure integer arithmetic in uint64 -> float64 -> float32;
no libm calls, so the values are identical in the build container and on the GPU box.
"""
from __future__ import annotations

import zlib

import numpy as np
import torch


def _hash_uniform(n: int, key: int) -> np.ndarray:
    """n values in [-0.5, 0.5), deterministic function of (index, key)."""
    M = np.uint64(0xFFFFFFFF)
    x = (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B1) + np.uint64(key & 0xFFFFFFFF)) & M
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x85EBCA77)) & M
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE3D)) & M
    x ^= x >> np.uint64(16)
    return x.astype(np.float64) / 4294967296.0 - 0.5


def _key(name: str) -> int:
    return zlib.crc32(name.encode())


def tensor_for(name: str, shape, kind: str | None = None) -> torch.Tensor:
    """Deterministic fp32 tensor for a parameter called `name`.

    kind: 'weight' (2-D, scaled 1.6/sqrt(fan_in)), 'ln_w' (1 + 0.2u), 'bias' (0.1u), 'embed' (0.4u),
    'input' (2u).  Inferred from the name when None."""
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if shape else 1
    u = _hash_uniform(n, _key(name))
    if kind is None:
        leaf = name.split(".")[-1]
        parent = name.split(".")[-2] if "." in name else ""
        if leaf == "weight" and len(shape) == 1:
            kind = "ln_w"
        elif leaf == "bias":
            kind = "bias"
        elif leaf in ("modal_token", "modality_embedding") or parent in ("wpe", "position_embeddings"):
            kind = "embed"
        else:
            kind = "weight"
    if kind == "weight":
        # nn.Linear is [out,in]; HF Conv1D is [in,out]; use the geometric mean so both are sane
        fan = float(np.sqrt(shape[0] * shape[1])) if len(shape) == 2 else float(shape[-1])
        v = u * (3.2 / np.sqrt(fan))
    elif kind == "ln_w":
        v = 1.0 + 0.4 * u
    elif kind == "bias":
        v = 0.2 * u
    elif kind == "embed":
        v = 0.8 * u
    elif kind == "input":
        v = 4.0 * u
    else:
        raise ValueError(kind)
    return torch.from_numpy(v.astype(np.float32).reshape(shape))


def fill_state(shapes: dict) -> dict:
    """shapes: {state_dict name: shape} -> {name: tensor}."""
    return {k: tensor_for(k, s) for k, s in shapes.items()}


def inputs_for(tag: str, modal_dims: dict, B: int, T: int) -> dict:
    """{mod: (B,T,C,1,1,1)} in the loader layout."""
    return {m: tensor_for(f"{tag}.input.{m}", (B, T, C, 1, 1, 1), "input") for m, C in modal_dims.items()}


def labels_for(tag: str, B: int, T: int, num_classes: int, ignore_frac: float = 0.25):
    u = _hash_uniform(B, _key(tag + ".target"))
    tgt = torch.from_numpy(((u + 0.5) * num_classes).astype(np.int64).clip(0, num_classes - 1))
    u2 = _hash_uniform(B * T, _key(tag + ".target_subclips"))
    sub = ((u2 + 0.5) * num_classes).astype(np.int64).clip(0, num_classes - 1)
    u3 = _hash_uniform(B * T, _key(tag + ".ignore"))
    sub[(u3 + 0.5) < ignore_frac] = -1
    return tgt, torch.from_numpy(sub.reshape(B, T, 1))


def eval_inputs(N=48, A=30, V=7, Nn=11, seed=0):
    """closed-form inputs of the eval-path fixture: action logits, one-hot action -> verb / noun maps, labels"""
    logits = (8.0 * _hash_uniform(N * A, 0xE7A1 + seed)).astype(np.float32).reshape(N, A)
    a = np.arange(A)
    verb_of, noun_of = (a * 3 + 1) % V, (a * 5 + 2) % Nn
    mv = np.zeros((A, V), np.float32)
    mv[a, verb_of] = 1
    mn = np.zeros((A, Nn), np.float32)
    mn[a, noun_of] = 1
    labels = (np.arange(N) * 7 + 3) % A
    labels[::4] = logits[::4].argmax(1)          # some clips are predicted right
    return logits, mv, mn, labels, verb_of[labels], noun_of[labels]




def unseen_tail_tables(N=48, A=30):
    """closed-form EPIC-100 bookkeeping of the unseen / tail fixture: one narration id per clip, the four RULSTM id tables
    (unseen participants; tail verbs / nouns / actions: each also holds ids that are not in the split) and many-shot subsets"""
    ids = np.asarray([f"P{1 + i % 9:02d}_{100 + i % 4}_{i}" for i in range(N)])
    i = np.arange(N)
    tables = {"validation_unseen_participants_ids.csv": list(ids[i % 5 == 0]) + ["P35_105_7"],
              "validation_tail_verbs_ids.csv": list(ids[i % 3 != 0]),
              "validation_tail_nouns_ids.csv": list(ids[(i % 4 == 1) | (i > 40)]) + ["P01_101_999"],
              "validation_tail_actions_ids.csv": list(ids[i % 2 == 1])}
    manyshot = {"verb": {f"v{c}": c for c in (0, 2, 3)}, "noun": {f"n{c}": c for c in (1, 4, 5, 9)},
                "action": {f"a{c}": c for c in range(0, A, 3)}}
    return ids, tables, manyshot


def reader_stores(C_rgb=12, C_audio=6):
    """closed-form RULSTM-style feature stores for the reader fixture: key "<video>_frame_{:010d}.jpg" -> float32 bytes.
    'rgb' store (30 fps ids): frames 3.. of P01_101 stored except every 7th and a hole of 12 (beyond the 9-frame search);
    'audio' store (indexed in the ORIGINAL video's 50 fps for the EPIC-100 name P01_101): every second frame stored."""
    vid = "P01_101"
    rgb, audio = {}, {}
    for f in range(3, 140):
        if f % 7 == 0 or 60 <= f < 72:
            continue
        rgb[f"{vid}_frame_{f:010d}.jpg".encode()] = (_hash_uniform(C_rgb, 0xA000 + f)).astype(np.float32).tobytes()
    for f in range(2, 260, 2):
        audio[f"{vid}_frame_{f:010d}.jpg".encode()] = (_hash_uniform(C_audio, 0xB000 + f)).astype(np.float32).tobytes()
    queries = [(0.05, 0.9), (1.0, 2.45), (1.9, 2.5), (2.0, 4.0)]      # (start_sec, end_sec); the first clamps ids < 1
    return vid, {"rgb": rgb, "audio": audio}, queries

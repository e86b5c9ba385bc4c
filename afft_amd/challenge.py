"""GPU side of the reference's ``challenge.marginalize_verb_noun`` (challenge.py:196-210): action logits -> softmax ->
verb and noun scores by the class-mapping matrices, as one row-softmax kernel and two exact-fp32 MFMA GEMMs on tensors
that are already on the device (the reference does this in numpy on host copies), followed by the reference's accuracy
bookkeeping (top-1 / top-5 / mean top-5 recall per verb, noun and action: challenge.py:94-106,161-193, common/utils.py:19-56)
on ONE host copy of the three score matrices.  Pinned by tests/golden/m0_marginalize.npz, produced by the reference's own
``marginalize_verb_noun`` on a stub dataset object."""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
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


EPIC100_VERSION = 0.2       # datasets/epic_kitchens.py: the dataset object's `version` for EPIC-KITCHENS-100


def topk_accuracy(scores: np.ndarray, labels: np.ndarray, ks, selected_class=None):
    """common/utils.py:19-42: share of rows whose label is among the k best scores.  Ties rank as in the reference's
    `scores.argsort()[:, ::-1]`: among equal scores the HIGHER class index comes first."""
    if selected_class is not None:
        keep = labels == selected_class
        scores, labels = scores[keep], labels[keep]
    kmax = int(max(ks))
    order = np.argsort(scores, axis=1, kind="stable")[:, ::-1][:, :kmax]
    hit = order == labels.reshape(-1, 1)
    return [hit[:, :k].any(axis=1).mean() for k in ks]


def topk_recall(scores: np.ndarray, labels: np.ndarray, k: int = 5, classes=None):
    """common/utils.py:45-56: mean over the classes present in `labels` of their top-k accuracy"""
    present = np.unique(labels)
    classes = present if classes is None else np.intersect1d(classes, present)
    return sum(topk_accuracy(scores, labels, ks=(k,), selected_class=c)[0] for c in classes) / len(classes)


def compute_accuracy(predictions: np.ndarray, labels: np.ndarray, classes=None):
    """challenge.py:94-106: (top-1, top-5, mean top-5 recall) in percent; classes: optional {name: class id} subset"""
    if classes is not None:
        classes = list(classes.values())
    top1, top5 = topk_accuracy(predictions, labels, ks=(1, 5))
    return top1 * 100, top5 * 100, topk_recall(predictions, labels, k=5, classes=classes) * 100


def _read_id_column(path: str) -> np.ndarray:
    """the RULSTM id tables are header-less one-column CSV files (one narration id per line)"""
    with open(path, "r", encoding="utf-8") as fh:
        return np.asarray([line.strip() for line in fh if line.strip()])


def epic100_unseen_tail_eval(probs, dataset) -> Dict[str, float]:
    """challenge.py:109-158: mean top-5 recall on the validation segments of unseen participants and on the segments whose
    verb / noun / action is a tail class.  The four id tables are read from dataset.rulstm_annotation_dir
    (validation_unseen_participants_ids.csv, validation_tail_{verbs,nouns,actions}_ids.csv) and matched against
    dataset.df.narration_id; the verb table selects rows for the verb scores, and so on; the unseen table for all three."""
    import os.path as osp  # noqa: PLC0415
    narration = np.asarray(dataset.df.narration_id.values).astype(str)
    res = {}
    tables = {"tail": {k: f"validation_tail_{k}s_ids.csv" for k in ("verb", "noun", "action")},
              "unseen": dict.fromkeys(("verb", "noun", "action"), "validation_unseen_participants_ids.csv")}
    for split, files in tables.items():
        for i, key in enumerate(("verb", "noun", "action")):
            rows = np.isin(narration, _read_id_column(osp.join(dataset.rulstm_annotation_dir, files[key])))
            labels = getattr(dataset.df, f"{key}_class").values[rows]
            res[f"{key[0]}mt5r_{split}"] = compute_accuracy(np.asarray(probs[i])[rows], labels)[2]
    return res


def compute_accuracies_epic(probs, dataset, compute_manyshot_unseen_tail: bool = False) -> Dict[str, float]:
    """challenge.py:161-193: probs = [verb, noun, action] score matrices; dataset.df carries verb_class / noun_class /
    action_class, dataset.classes_manyshot the many-shot subsets; with compute_manyshot_unseen_tail on an EPIC-100 dataset
    the unseen / tail recalls of :109-158 are added."""
    assert len(probs) == 3, 'Probs should contain probs for verb, noun and action'
    many = dataset.classes_manyshot
    res = {}
    for short, key, p in (("v", "verb", probs[0]), ("n", "noun", probs[1]), ("a", "action", probs[2])):
        labels = getattr(dataset.df, f"{key}_class").values
        p = np.asarray(p)
        res[f"{short}top1"], res[f"{short}top5"], res[f"{short}mt5r"] = compute_accuracy(p, labels)
        res[f"{short}mt5r_ms"] = float("nan")
        if key in many and compute_manyshot_unseen_tail:
            res[f"{short}mt5r_ms"] = compute_accuracy(p, labels, classes=many[key])[2]
    if compute_manyshot_unseen_tail and getattr(dataset, "version", None) == EPIC100_VERSION:
        res.update(epic100_unseen_tail_eval(probs, dataset))
    return res


def marginalize_verb_noun(res_action, dataset, to_prob: bool = True, compute_manyshot_unseen_tail: bool = False):
    """challenge.py:196-210 with the same signature and return value: (accuracies, [verb, noun, action] as numpy arrays).
    res_action: action logits, a device tensor (kept on the device for the softmax and the two mapping GEMMs) or anything
    torch.as_tensor accepts (moved to cuda:0)."""
    x = torch.as_tensor(res_action)
    if not x.is_cuda:
        x = x.cuda()
    verb, noun, action = marginalize_scores(x.float(), dataset.class_mappings, to_prob=to_prob)
    scores = [t.detach().cpu().numpy() for t in (verb, noun, action.reshape(-1, action.shape[-1]))]
    return compute_accuracies_epic(scores, dataset, compute_manyshot_unseen_tail), scores


LOGITS_DIR = 'logits'        # challenge.py:28
PREFIX_H5 = 'test'           # challenge.py:32


def gen_load_resfiles(resdir: str):
    """challenge.py:79-91: every '<PREFIX_H5>*h5' logits file of a run directory as {leaf key: array} (through h5py when it is
    installed, else through afft_amd.h5lite, which reads the same files)."""
    import glob  # noqa: PLC0415
    import os.path as osp  # noqa: PLC0415
    from .evaluate import load_logits  # noqa: PLC0415
    resfiles = sorted(glob.glob(osp.join(resdir, PREFIX_H5 + '*h5')))
    if len(resfiles) == 0:
        raise ValueError(f'Didnt find any resfiles in {resdir}')
    for resfile in resfiles:
        yield load_logits(resfile)


def get_epic_marginalize_verb_noun(resdir: str, dataset):
    """challenge.py:213-222: verb / noun scores from the action logits stored in a run directory."""
    res = next(gen_load_resfiles(resdir))
    res_action = None
    for key, val in res.items():
        if key.startswith('logits/action'):
            res_action = val
    assert res_action is not None, 'Can not find logits/action in h5.'
    return marginalize_verb_noun(res_action, dataset)

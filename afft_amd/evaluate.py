"""Inference / evaluation side of the path: the forward-only loops of the reference's ``test.py``
(``evaluate`` :66-102, ``save_logits`` :33-63) without its per-batch host round trips.

The reference copies every batch of logits to the host (``.detach().cpu().numpy()``, a device sync per batch),
concatenates there and marginalises verbs / nouns in numpy.  Here the logits of the branch the reference keeps (the only
modality, or 'all-fused') stay on the device, are concatenated once, marginalised by ``afft_amd.challenge.
marginalize_scores`` (row softmax + two fp32 MFMA GEMMs) and leave the device in ONE copy per result.  The accuracy
bookkeeping over the dataset's annotations (``challenge.compute_accuracies_epic``) stays with the caller.
"""
from __future__ import annotations

import logging
import os
from typing import Dict, Iterable, Optional, Tuple

import numpy as np
import torch

from . import runtime as rt
from .challenge import marginalize_scores

LOGITS_KEY = 'logits/action'


def _eval_kwargs():
    return dict(mixup_fn=None, target=None, target_subclips=None, target_subclips_ignore_index=None)


def _branch(outputs) -> Tuple[str, torch.Tensor]:
    """the entry test.py keeps: the single modality / early-fusion branch, else 'all-fused' (test.py:48-58)"""
    heads = outputs[LOGITS_KEY]
    if len(heads) == 1:
        modk = next(iter(heads.keys()))
    else:
        modk = 'all-fused'
        logging.info('This model consists of multiple branches. Saving fusion branch "%s" only ...', modk)
    return f'{LOGITS_KEY}_{modk}', heads[modk][:, 0, :]


@torch.no_grad()
def collect_logits(model, data_loader: Iterable, device, precision: Optional[str] = None) -> Tuple[str, torch.Tensor]:
    """(key, fp32 [N, classes] ON THE DEVICE) over the whole loader; batches are ``(data, timings)`` pairs with
    ``data['data_dict']`` as the reference's loader yields them.  precision: run these forward passes in another precision of
    the library and restore the current one -- 'fp16x2' gives logits within 1e-3 of the reference's fp32 arithmetic (6.4e-4 on
    cfg2) where the training mode 'bf16' is at 7.8e-3, at 2.3x the bf16 forward's time."""
    model.eval()
    key, parts = None, []
    for data in data_loader:
        data, _ = data
        feats = {mod: t.to(device, non_blocking=True) for mod, t in data["data_dict"].items()}
        with rt.precision_scope(precision):
            outputs, _ = model(feats, **_eval_kwargs())
        key, lg = _branch(outputs)
        parts.append(lg.detach())
    assert parts, "empty data loader"
    return key, torch.cat(parts, dim=0)


@torch.no_grad()
def evaluate_scores(model, class_mappings, data_loader: Iterable, device, to_prob: bool = True,
                    precision: Optional[str] = None) -> Dict[str, np.ndarray]:
    """verb / noun / action score matrices of test.py:evaluate -> challenge.marginalize_verb_noun (:196-210), computed on
    the device; returns numpy arrays (one device-to-host copy each)."""
    _, logits = collect_logits(model, data_loader, device, precision=precision)
    verb, noun, action = marginalize_scores(logits, class_mappings, to_prob=to_prob)
    return {"verb": verb.cpu().numpy(), "noun": noun.cpu().numpy(), "action": action.cpu().numpy()}


def store_append(endpoints: Dict[str, np.ndarray], output_dir: str, save_file_name: str) -> str:
    """test.py:20-31: append-able, gzip-9, chunked HDF5 datasets keyed like 'logits/action_<modk>', first dimension unlimited.
    Through h5py when it is installed, otherwise through afft_amd.h5lite, which writes the same file format itself (h5py reads and
    extends its files, tests/test_h5lite_cpu.py)."""
    os.makedirs(output_dir, exist_ok=True)
    path = os.path.join(output_dir, save_file_name)
    try:
        import h5py  # noqa: PLC0415
    except ImportError:
        h5py = None
    if h5py is not None:
        with h5py.File(path, 'a') as fout:
            for key, val in endpoints.items():
                if key not in fout:
                    fout.create_dataset(key, data=val, compression='gzip', compression_opts=9, chunks=True,
                                        maxshape=(None,) + val.shape[1:])
                else:
                    fout[key].resize((fout[key].shape[0] + val.shape[0],) + val.shape[1:])
                    fout[key][-val.shape[0]:, ...] = val
        return path
    from . import h5lite  # noqa: PLC0415
    h5lite.append(path, {k: np.asarray(v) for k, v in endpoints.items()})
    return path


def load_logits(path: str, key: Optional[str] = None):
    """The stored logits back (what challenge.py does with h5py for ensembling): one dataset, or {key: array} for all."""
    try:
        import h5py  # noqa: PLC0415
    except ImportError:
        from . import h5lite  # noqa: PLC0415
        return h5lite.read(path, key)
    with h5py.File(path, 'r') as fin:
        if key is not None:
            return fin[key][...]
        out = {}
        fin.visititems(lambda name, obj: out.__setitem__(name, obj[...]) if isinstance(obj, h5py.Dataset) else None)
        return out


@torch.no_grad()
def save_logits(model, data_loader: Iterable, device, logger=None, save_dir: Optional[str] = None,
                save_file_name: Optional[str] = None, precision: Optional[str] = None) -> str:
    """test.py:33-63: logits of the kept branch for ensembling / analysis; one host copy for the whole loader."""
    key, logits = collect_logits(model, data_loader, device, precision=precision)
    path = store_append({key: logits.cpu().numpy()}, save_dir, save_file_name)
    if logger is not None:
        logger.info(f'Saved logits {[key]} as {save_file_name} to {save_dir}.')
    return path

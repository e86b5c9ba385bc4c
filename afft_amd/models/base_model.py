"""MI355X mirror of the reference's ``models/base_model.py`` (BaseModel, :15-119): the API shell the training
loop talks to.  Contract kept from the reference: constructor ``(model_cfg, num_classes, class_mappings)``,
``backbone`` ModuleDict + ``future_predictor`` attribute names (state_dict keys), ``cls_map_*`` buffers,
``forward(video_data, *, mixup_fn, target, target_subclips, target_subclips_ignore_index)`` returning
``({key: {modality: tensor}}, {'target', 'target_subclips', 'target_subclips_ignore_index'})``, 6-D single-crop and
7-D multi-crop inputs (outputs averaged over the crops, attention maps of the first crop).  The body is organised
differently: crops are enumerated up front as a list of per-crop feature dicts, the loader layout
(B, #clips, C, 1, 1, 1) is reduced to (B, T, C) per modality by one helper, and the crop average is a single
stack-and-mean per output.  ``future_predictor`` is this package's CMFPEarly, whose arithmetic runs in HIP kernels."""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.nn as nn

from .._hydra_compat import instantiate

CLS_MAP_PREFIX = 'cls_map_'
PAST_LOGITS_PREFIX = 'past_'


def _clip_features(feats: torch.Tensor) -> torch.Tensor:
    """backbone output (B, #clips, C, T', H, W) -> (B, #clips * T', C): spatial mean, time before channels, clips and
    their frames flattened into one time axis (models/base_model.py:35-42).  Pre-extracted features have H = W = 1,
    where the mean is a view."""
    if feats.ndim >= 5 and feats.shape[-2:] == (1, 1):
        pooled = feats[..., 0, 0]
    else:
        pooled = feats.mean(dim=(-1, -2))
    pooled = pooled.transpose(-1, -2)             # (B, #clips, T', C)
    return pooled.flatten(1, 2) if pooled.ndim == 4 else pooled


def _per_crop_inputs(video_data: Dict[str, torch.Tensor]) -> List[Dict[str, torch.Tensor]]:
    """{mod: 6-D or 7-D tensor} -> one {mod: 6-D tensor} per crop.  A modality with fewer crops than the widest one is
    cycled (whole repetitions only, like the reference's list replication followed by zip, models/base_model.py:84-89)."""
    views = {}
    for mod in sorted(video_data):
        x = video_data[mod]
        if x.ndim == 6:
            views[mod] = (x,)
        elif x.ndim == 7:
            views[mod] = x.unbind(dim=2)
        else:
            raise NotImplementedError('Unsupported size %s' % (tuple(x.shape),))
    widest = max(len(v) for v in views.values())
    n_crops = min(len(v) * (widest // len(v)) for v in views.values())
    return [{mod: v[i % len(v)] for mod, v in views.items()} for i in range(n_crops)]


def _average_over_crops(per_crop: List[dict]) -> dict:
    first = per_crop[0]
    merged = {}
    for key, by_mod in first.items():
        if len(per_crop) == 1 or key == 'attentions':      # attention maps are reported for the first crop only
            merged[key] = dict(by_mod)
        else:
            merged[key] = {m: torch.stack([o[key][m] for o in per_crop]).mean(dim=0) for m in by_mod}
    return merged


class BaseModel(nn.Module):
    def __init__(self, model_cfg, num_classes: Dict[str, int],
                 class_mappings: Dict[Tuple[str, str], torch.FloatTensor]):
        super().__init__()
        self.backbone = nn.ModuleDict({mod: instantiate(conf) for mod, conf in model_cfg.common.backbones.items()})
        self.future_predictor = instantiate(model_cfg.CMFP, model_cfg=model_cfg, num_classes=num_classes,
                                            _recursive_=False)
        for (src, dst), mapping in class_mappings.items():
            self.register_buffer(f'{CLS_MAP_PREFIX}{src}_{dst}', mapping)

    def forward_singlecrop(self, data_dict, **kwargs):
        """One crop: backbones (Identity for pre-extracted features), (B, T, C) layout, optional MixUp on the features
        and labels, the fusion + anticipation model.  Returns (outputs, targets as the loss must see them)."""
        feats = {mod: _clip_features(self.backbone[mod](x)) for mod, x in data_dict.items()}
        labels = {k: kwargs[k] for k in ('target', 'target_subclips', 'target_subclips_ignore_index')}
        mixup_fn = kwargs['mixup_fn']
        if mixup_fn is not None:
            feats, labels['target'], labels['target_subclips'], labels['target_subclips_ignore_index'] = \
                mixup_fn(feats, labels['target'], labels['target_subclips'])
        return self.future_predictor(feats), labels

    def forward(self, video_data, *args, **kwargs):
        """video_data: {mod: (B, #clips, C, T, H, W) or (B, #clips, #crops, C, T, H, W)}"""
        results = [self.forward_singlecrop(crop, *args, **kwargs) for crop in _per_crop_inputs(video_data)]
        # MixUp only happens in training, where there is a single crop: the first crop's targets stand for all
        return _average_over_crops([out for out, _ in results]), results[0][1]

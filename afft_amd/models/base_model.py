"""MI355X mirror of the reference's ``models/base_model.py`` (BaseModel, :15-119): the API shell the
training loop talks to.  Identity backbones, (B, #clips, C, 1, 1, 1) -> (B, T, C) feature layout, the
optional MixUp hook, multi-crop averaging and the nested ``{key: {modality: tensor}}`` output dict are kept
as in the reference; ``future_predictor`` is this package's CMFPEarly, whose arithmetic runs in HIP kernels."""
from __future__ import annotations

from itertools import repeat
from typing import Dict, Tuple

import torch
import torch.nn as nn

from .._hydra_compat import instantiate

CLS_MAP_PREFIX = 'cls_map_'
PAST_LOGITS_PREFIX = 'past_'


class BaseModel(nn.Module):
    def __init__(self, model_cfg, num_classes: Dict[str, int],
                 class_mappings: Dict[Tuple[str, str], torch.FloatTensor]):
        super().__init__()
        self.backbone = nn.ModuleDict()
        for mod, backbone_conf in model_cfg.common.backbones.items():
            self.backbone[mod] = instantiate(backbone_conf)
        self.future_predictor = instantiate(model_cfg.CMFP, model_cfg=model_cfg, num_classes=num_classes,
                                            _recursive_=False)
        for (src, dst), mapping in class_mappings.items():
            self.register_buffer(f'{CLS_MAP_PREFIX}{src}_{dst}', mapping)

    def forward_singlecrop(self, data_dict, **kwargs):
        feats_past = {}
        for mod, data in data_dict.items():
            feats = self.backbone[mod](data)
            if feats.ndim >= 5 and feats.shape[-1] == 1 and feats.shape[-2] == 1:
                feats = feats[..., 0, 0]            # spatial mean over a 1x1 map is a view
            else:
                feats = torch.mean(feats, [-1, -2])
            feats = feats.permute((0, 1, 3, 2))     # B x clips x T x C
            if feats.ndim == 4:
                feats = torch.flatten(feats, 1, 2)  # B x T x C
            feats_past[mod] = feats

        target = kwargs['target']
        target_subclips = kwargs['target_subclips']
        target_subclips_ignore_index = kwargs['target_subclips_ignore_index']
        if kwargs['mixup_fn'] is not None:
            mixup_fn = kwargs['mixup_fn']
            feats_past, target, target_subclips, target_subclips_ignore_index = \
                mixup_fn(feats_past, target, target_subclips)

        outputs = self.future_predictor(feats_past)
        outputs_target = {'target': target, 'target_subclips': target_subclips,
                          'target_subclips_ignore_index': target_subclips_ignore_index}
        return outputs, outputs_target

    def forward(self, video_data, *args, **kwargs):
        """video_data: {mod: (B, #clips, C, T, H, W) or (B, #clips, #crops, C, T, H, W)}"""
        video_data = dict(video_data)
        for mod, data in video_data.items():
            if data.ndim == 6:
                video_data[mod] = [data]
            elif data.ndim == 7 and data.size(2) == 1:
                video_data[mod] = [data.squeeze(2)]
            elif data.ndim == 7:
                video_data[mod] = torch.unbind(data, dim=2)
            else:
                raise NotImplementedError('Unsupported size %s' % (tuple(data.shape),))

        all_mods = sorted(list(video_data.keys()))
        all_data = [video_data[mod] for mod in all_mods]
        num_crops = max([len(sl) for sl in all_data])
        all_data = [list(sl) * (num_crops // len(sl)) for sl in all_data]
        all_crops = list(zip(*all_data))
        crops = [{m: c for m, c in zip(mods, cr)} for mods, cr in zip(repeat(all_mods), all_crops)]
        feats = [self.forward_singlecrop(el, *args, **kwargs) for el in crops]
        output_targets = feats[0][1]  # mixup only happens in training, where there is a single crop

        if len(feats) == 1:
            merged = {}
            for key, val in feats[0][0].items():
                merged[key] = dict(val)
            return merged, output_targets

        feats_merged = {}
        for out_dict, _ in feats:
            for key in out_dict:
                if key not in feats_merged:
                    feats_merged[key] = {k: [v] for k, v in out_dict[key].items()}
                else:
                    for k, v in feats_merged[key].items():
                        v.append(out_dict[key][k])
        for out_key in feats_merged:
            if out_key == 'attentions':
                feats_merged[out_key] = {k: el[0] for k, el in feats_merged[out_key].items()}
                continue
            feats_merged[out_key] = {k: torch.mean(torch.stack(el, dim=0), dim=0)
                                     for k, el in feats_merged[out_key].items()}
        return feats_merged, output_targets

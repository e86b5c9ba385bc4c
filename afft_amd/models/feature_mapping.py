"""MI355X mirror of the reference's ``models/feature_mapping.py``: ``Linear`` (models/feature_mapping.py:54-78),
the per-modality projection to ``common_dim`` used by every BASELINE config (bias-free nn.Linear, or Identity
when in == out under sparse_mapping).  GatedLinear / NonLinear are unused by expts/01 and expts/04
 and are not provided."""
from __future__ import annotations

from functools import partial

import torch
from torch import nn as nn

from .. import functional as F_


class _HipLinear(nn.Linear):
    """nn.Linear whose forward is the MFMA GEMM (any leading dims)."""

    def forward(self, x):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        if x2.dtype != torch.float32:
            x2 = x2.float()
        y = F_.Linear.apply(x2, self.weight, self.bias)
        return y.view(*shp[:-1], self.out_features)


class _HipLayerNorm(nn.LayerNorm):
    def forward(self, x):
        shp = x.shape
        y = F_.LayerNormRows.apply(x.reshape(-1, shp[-1]).float().contiguous(), self.weight, self.bias, self.eps, 1)
        return y.view(shp)


class Linear(nn.Module):
    """Implements the linear feature mapping layer"""

    def __init__(self, in_features, out_features, use_layernorm: bool = False, sparse_mapping=True):
        super().__init__()
        if sparse_mapping:
            layers = [_HipLinear(in_features, out_features, bias=False)
                      if in_features != out_features else nn.Identity()]
        else:
            layers = [_HipLinear(in_features, out_features, bias=False)]
        if use_layernorm:
            layers.append(_HipLayerNorm(out_features, eps=1e-6))
        self.mapping = nn.Sequential(*layers)
        self.use_layernorm = use_layernorm
        self.sparse_mapping = sparse_mapping

    def forward(self, x):
        return self.mapping(x)

    def __str__(self):
        return f'Linear mapping layer with use_layernorm: {self.use_layernorm}, ' \
               f'and sparse_mapping: {self.sparse_mapping}'

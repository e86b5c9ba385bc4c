"""MI355X mirror of the reference's ``models/feature_mapping.py``: ``Linear`` (models/feature_mapping.py:54-78),
the per-modality projection to ``common_dim`` used by every BASELINE config (bias-free nn.Linear, or Identity
when in == out under sparse_mapping), plus GatedLinear (:33-51) and NonLinear (:91-112) on the same GEMM kernel
with the activation / gate fused into its epilogue (functional.LinearAct)."""
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


def _rows2d(x):
    shp = x.shape
    x2 = x.reshape(-1, shp[-1])
    return (x2 if x2.dtype == torch.float32 else x2.float()), shp


class ContextGating(nn.Module):
    """x * sigmoid(fc(x)) -- the reference's glu(cat((x, fc(x)), 1), 1) (models/feature_mapping.py:22-31)."""

    def __init__(self, dimension):
        super().__init__()
        self.fc = nn.Linear(dimension, dimension)

    def forward(self, x):
        x2, shp = _rows2d(x)
        return F_.LinearAct.apply(x2, self.fc.weight, self.fc.bias, "gate", x2, None).view(shp)


class GatedEmbeddingUnit(nn.Module):
    def __init__(self, input_dimension, output_dimension):
        super().__init__()
        self.fc = _HipLinear(input_dimension, output_dimension)
        self.cg = ContextGating(output_dimension)

    def forward(self, x):
        return self.cg(self.fc(x))


class GatedLinear(nn.Module):
    def __init__(self, in_features, out_features, use_layernorm: bool = True):
        super().__init__()
        tmp = [_HipLinear(in_features, out_features), ContextGating(out_features)]
        if use_layernorm:
            tmp.append(_HipLayerNorm(out_features, eps=1e-6))
        self.mapping = nn.Sequential(*tmp)
        self.use_layernorm = use_layernorm

    def forward(self, x):
        return self.mapping(x)

    def __str__(self):
        return f'Gated linear mapping layer with use_layernorm: {self.use_layernorm}'


def get_activation_layer(name):
    act_layers = {'relu': nn.ReLU(), 'gelu': nn.GELU(), 'none': nn.Identity()}
    assert name in act_layers.keys(), f'{name} is not supported in {list(act_layers.keys())}.'
    return act_layers[name]


class NonLinear(nn.Module):
    """Implements the non-linear feature mapping layer: Linear (+bias) -> relu | gelu | none (-> LayerNorm); the
    activation runs in the GEMM epilogue, `mapping.1` only keeps the reference's module / state_dict layout."""

    def __init__(self, in_features, out_features, use_layernorm: bool = False, activation='relu'):
        super().__init__()
        layers = [nn.Linear(in_features, out_features), get_activation_layer(activation)]
        if use_layernorm:
            layers.append(_HipLayerNorm(out_features, eps=1e-6))
        self.mapping = nn.Sequential(*layers)
        self.use_layernorm = use_layernorm
        self.activation = activation

    def forward(self, x):
        x2, shp = _rows2d(x)
        lin = self.mapping[0]
        y = F_.LinearAct.apply(x2, lin.weight, lin.bias, self.activation, None, None).view(*shp[:-1], lin.out_features)
        return self.mapping[2](y) if self.use_layernorm else y

    def __str__(self):
        return f'Nonlinear mapping layer with use_layernorm: {self.use_layernorm}, ' \
               f'and activation: {self.activation}'

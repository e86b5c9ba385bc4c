"""MI355X mirror of the reference's ``models/future_prediction.py``:

  CrossModalFusionPrediction (helpers)  <- models/future_prediction.py:19-186
  IndividualFuturePrediction            <- models/future_prediction.py:189-225
  CMFPEarly                             <- models/future_prediction.py:228-291
  CMFPScoreFusion                       <- models/future_prediction.py:294-351
  BaseFuturePredictor                   <- models/future_prediction.py:354-415  (incl. the fp_output_len > 1 roll-out)

The reference's temporal predictor is HuggingFace ``transformers.GPT2Model`` (pinned 4.18.0,
environment.yml:166).  Here the same network is built natively (no ``transformers`` import) with HF's
parameter names and layouts -- ``gpt_model.wpe.weight``, ``gpt_model.h.{i}.{ln_1,ln_2}``,
``.attn.c_attn/.attn.c_proj/.mlp.c_fc/.mlp.c_proj`` as Conv1D ``[in, out]`` weights, ``gpt_model.ln_f`` --
so reference checkpoints load by name.  GPT-2 semantics restated from HF modeling_gpt2.py: pre-LN blocks,
eps 1e-5, c_attn bias, causal softmax scaled by head_dim^-0.5, gelu_new, learned absolute positions.
"""
from __future__ import annotations

import abc
import logging
import math
import warnings
from functools import partial
from typing import Dict, List, Tuple

import torch
import torch.nn as nn
from torch import Tensor

from .. import dropout as D_
from .. import functional as F_
from .._hydra_compat import instantiate, is_dict_config

PAST_LOGITS_PREFIX = 'past_'


# --------------------------------------------------------------------------- native GPT-2 (HF names)
class Conv1D(nn.Module):
    """HF Conv1D: y = x @ weight + bias with weight stored [in, out]."""

    def __init__(self, nf, nx):
        super().__init__()
        self.nf = nf
        self.weight = nn.Parameter(torch.empty(nx, nf))
        self.bias = nn.Parameter(torch.zeros(nf))
        nn.init.normal_(self.weight, std=0.02)


class GPT2Attention(nn.Module):
    def __init__(self, n_embd, n_head, attn_pdrop, resid_pdrop):
        super().__init__()
        self.num_heads = n_head
        self.c_attn = Conv1D(3 * n_embd, n_embd)
        self.c_proj = Conv1D(n_embd, n_embd)
        self.attn_dropout = nn.Dropout(attn_pdrop)
        self.resid_dropout = nn.Dropout(resid_pdrop)


class GPT2MLP(nn.Module):
    def __init__(self, n_embd, resid_pdrop):
        super().__init__()
        self.c_fc = Conv1D(4 * n_embd, n_embd)
        self.c_proj = Conv1D(n_embd, 4 * n_embd)
        self.dropout = nn.Dropout(resid_pdrop)


class GPT2Block(nn.Module):
    def __init__(self, n_embd, n_head, attn_pdrop, resid_pdrop, eps=1e-5):
        super().__init__()
        self.ln_1 = nn.LayerNorm(n_embd, eps=eps)
        self.attn = GPT2Attention(n_embd, n_head, attn_pdrop, resid_pdrop)
        self.ln_2 = nn.LayerNorm(n_embd, eps=eps)
        self.mlp = GPT2MLP(n_embd, resid_pdrop)

    def forward_rows(self, h: Tensor, L: int) -> Tuple[Tensor, Tensor]:
        a, m = self.attn, self.mlp
        h, probs = F_.AttnSublayer.apply(h, self.ln_1.weight, self.ln_1.bias, a.c_attn.weight, a.c_attn.bias,
                                         a.c_proj.weight, a.c_proj.bias, L, a.num_heads, "causal", self.ln_1.eps,
                                         True, True, None,
                                         D_.cfg(self, attn=a.attn_dropout.p, out=a.resid_dropout.p))
        h = F_.MLPSublayer.apply(h, self.ln_2.weight, self.ln_2.bias, m.c_fc.weight, m.c_fc.bias, m.c_proj.weight,
                                 m.c_proj.bias, self.ln_2.eps, "tanh", True, True, D_.cfg(self, out=m.dropout.p))
        return h, probs


class GPT2Model(nn.Module):
    """The slice of HF GPT2Model the reference uses: inputs_embeds + wpe[position_ids] -> drop -> blocks -> ln_f."""

    def __init__(self, n_embd, n_layer, n_head, n_positions=1024, embd_pdrop=0.1, resid_pdrop=0.1, attn_pdrop=0.1,
                 layer_norm_epsilon=1e-5):
        super().__init__()
        self.embed_dim = n_embd
        self.wpe = nn.Embedding(n_positions, n_embd)
        self.drop = nn.Dropout(embd_pdrop)
        self.h = nn.ModuleList([GPT2Block(n_embd, n_head, attn_pdrop, resid_pdrop, layer_norm_epsilon)
                                for _ in range(n_layer)])
        self.ln_f = nn.LayerNorm(n_embd, eps=layer_norm_epsilon)
        nn.init.normal_(self.wpe.weight, std=0.02)
        for blk in self.h:  # HF _init_weights: residual projections scaled by 1/sqrt(2*n_layer)
            nn.init.normal_(blk.attn.c_proj.weight, std=0.02 / math.sqrt(2 * n_layer))
            nn.init.normal_(blk.mlp.c_proj.weight, std=0.02 / math.sqrt(2 * n_layer))

    def forward_rows(self, x2: Tensor, L: int, want_attn: bool = False):
        """x2 fp32 [B*L, D], positions 0..L-1 per clip. Returns (last_hidden_state rows, [probs per layer])."""
        h = F_.AddRowTable.apply(x2, self.wpe.weight, L, 0)
        if self.training and self.drop.p > 0:
            h = F_.ElementDropout.apply(h, D_.elementwise(self.drop.p))
        attns = []
        for blk in self.h:
            h, probs = blk.forward_rows(h, L)
            if want_attn:
                attns.append(probs)
        h = F_.LayerNormRows.apply(h, self.ln_f.weight, self.ln_f.bias, self.ln_f.eps, 1)
        return h, attns


class BaseFuturePredictor(nn.Module):
    """future predictor for single modality (GPT-2 style causal transformer over frames)"""

    def __init__(self, in_features, inter_dim=2048, n_layer=6, n_head=4, embd_pdrop=0.1, resid_pdrop=0.1,
                 attn_pdrop=0.1, output_attentions=False, dimension_mapping=False):
        super().__init__()
        self.in_features = in_features
        self.output_attentions = output_attentions
        if dimension_mapping:
            warnings.warn('Using dimension mapping inside GPT2 is deprecated.')
        self.encoder = nn.Linear(in_features, inter_dim, bias=False) if dimension_mapping else nn.Identity()
        self.decoder = nn.Linear(inter_dim, in_features, bias=False) if dimension_mapping else nn.Identity()
        self.gpt_model = GPT2Model(n_embd=inter_dim, n_layer=n_layer, n_head=n_head, embd_pdrop=embd_pdrop,
                                   resid_pdrop=resid_pdrop, attn_pdrop=attn_pdrop)

    def _map(self, lin, x: Tensor) -> Tensor:
        if isinstance(lin, nn.Identity):
            return x
        B, T, C = x.shape
        return F_.Linear.apply(x.reshape(B * T, C), lin.weight, None).view(B, T, -1)

    def forward(self, feats: torch.Tensor, output_len: int = 1) -> Tuple[torch.Tensor, Dict]:
        """feats (B, T, C) -> (B, T + output_len - 1, C), {gpt2_att_i: (B, layers, heads, L, L)}.

        output_len > 1: the reference rolls out with HF's KV cache, feeding the last hidden state back as the
        next input embedding at the next position (future_prediction.py:395-412).  A causal model's earlier
        positions do not change when a token is appended, so the roll-out is computed by re-running the
        extended sequence and keeping its last frame -- the same arithmetic without a cache."""
        addl_endpoints = {}
        feats = self._map(self.encoder, feats)
        B, T, D = feats.shape
        seq = feats if feats.dtype == torch.float32 else feats.float()
        outs: List[Tensor] = []
        for output_id in range(output_len):
            L = seq.shape[1]
            h, attns = self.gpt_model.forward_rows(seq.reshape(B * L, D).contiguous(), L, self.output_attentions)
            h = h.view(B, L, D)
            if self.output_attentions:
                a = torch.stack([p.view(B, -1, L, L) for p in attns]).transpose(0, 1)
                addl_endpoints[f'gpt2_att_{output_id}'] = a if output_id == 0 else a[:, :, :, -1:, :]
            new = h if output_id == 0 else h[:, -1:, :]
            outs.append(self._map(self.decoder, new))
            if output_id + 1 < output_len:
                seq = torch.cat([seq, h[:, -1:, :]], dim=1)
        return torch.cat(outs, dim=1), addl_endpoints


# --------------------------------------------------------------------------- cross-modal fusion + prediction
class CrossModalFusionPrediction(nn.Module, metaclass=abc.ABCMeta):
    """base class cross modality future predictor"""

    def __init__(self, model_cfg, num_classes, instantiate_: bool = True):
        super().__init__()
        assert is_dict_config(model_cfg.modal_dims), 'cfg.model.modal_dims must be a Dict!'
        self.cfg = model_cfg
        self.num_classes = num_classes
        self.latent_dim = model_cfg.common.in_features
        self.fp_inter_dim = model_cfg.common.fp_inter_dim
        self.modality_dims = model_cfg.modal_dims
        self.common_predictor = model_cfg.common.share_predictors
        self.common_classifier = model_cfg.common.share_classifiers
        self.modality_cls = model_cfg.common.modality_cls
        self.fusion_cls = model_cfg.common.fusion_cls
        if instantiate_:
            self.mapping = self._init_mapping_layer()
            self.fuser = self._init_fuser(model_cfg)
            self.future_predictor = self._init_future_predictor(model_cfg, self.common_predictor)
        self.classifiers = self._init_classifiers(self.latent_dim, self.modality_dims, self.num_classes,
                                                  self.common_classifier, self.cfg.dropout, self.modality_cls,
                                                  self.fusion_cls)

    def _init_mapping_layer(self):
        mapping_layer = nn.ModuleDict()
        for mod in self.modality_dims.keys():
            mapping_layer[mod] = instantiate(self.cfg.mapping, in_features=self.modality_dims[mod],
                                             out_features=self.latent_dim)
            logging.info(f'Using {mapping_layer[mod]} for {mod}')
        return mapping_layer

    @staticmethod
    def _init_fuser(model_cfg):
        return instantiate(model_cfg.fuser, _recursive_=False)

    @staticmethod
    def _init_dimension_encoder(modality_dims, inter_dim, latent_dim):
        """replaces the encoder inside gpt2, enabling modality specific dimension encoding"""
        del latent_dim
        return nn.ModuleDict({modk: (nn.Linear(mod_dim, inter_dim, bias=False) if mod_dim != inter_dim
                                     else nn.Identity()) for modk, mod_dim in modality_dims.items()})

    @staticmethod
    def _init_dimension_decoder(modality_dims, inter_dim, latent_dim):
        """replaces the decoder inside gpt2, enabling modality specific dimension decoding"""
        del latent_dim
        return nn.ModuleDict({modk: (nn.Linear(inter_dim, mod_dim, bias=False) if mod_dim != inter_dim
                                     else nn.Identity()) for modk, mod_dim in modality_dims.items()})

    @staticmethod
    def _project(lin, x: Tensor) -> Tensor:
        """bias-free nn.Linear (or Identity) of the dimension encoder / decoder on (B, T, C), as an MFMA GEMM"""
        if isinstance(lin, nn.Identity):
            return x
        B, T, C = x.shape
        return F_.Linear.apply(x.reshape(B * T, C), lin.weight, None).view(B, T, -1)

    def _init_future_predictor(self, model_cfg, common_predictor=False):
        self.dim_encoder = self._init_dimension_encoder(self.modality_dims, self.fp_inter_dim, self.latent_dim)
        self.dim_decoder = self._init_dimension_decoder(self.modality_dims, self.fp_inter_dim, self.latent_dim)
        if common_predictor:  # a common future predictor, features are mapped
            return instantiate(model_cfg.future_predictor, in_features=self.fp_inter_dim, dimension_mapping=False,
                               _recursive_=False)
        return nn.ModuleDict({modk: instantiate(model_cfg.future_predictor, in_features=self.fp_inter_dim,
                                                dimension_mapping=False, _recursive_=False)
                              for modk in model_cfg.modal_dims.keys()})

    def _predict_unimodal(self, z: Dict[str, Tensor]):
        """per-modality future prediction (models/future_prediction.py:204-217, :314-327)"""
        z_hat, attentions = {}, {}
        for modk, z_unimod in z.items():
            z_enc = self._project(self.dim_encoder[modk], z_unimod if z_unimod.dtype == torch.float32 else z_unimod.float())
            fp = self.future_predictor if self.common_predictor else self.future_predictor[modk]
            z_hat_enc, atts = fp(z_enc, self.cfg.common.fp_output_len)
            z_hat[modk] = self._project(self.dim_decoder[modk], z_hat_enc)
            attentions[modk] = atts
        return z_hat, attentions

    @staticmethod
    def _init_classifiers(latent_dim, modality_dims, num_classes, share_classifier, dropout, modality_cls,
                          fusion_cls):
        assert modality_cls or fusion_cls, 'Modality-level and / or fusion classification!'
        classifiers = nn.ModuleDict()
        for cls_type, cls_dim in num_classes.items():
            mod_classifiers = nn.ModuleDict()
            common_classifier = nn.Sequential(nn.Dropout(dropout), nn.Linear(latent_dim, cls_dim)
                                              ) if share_classifier else None
            if modality_cls:
                for modk, mod_dim in modality_dims.items():
                    mod_classifiers[modk] = nn.Sequential(nn.Dropout(dropout), nn.Linear(mod_dim, cls_dim)
                                                          ) if not common_classifier else common_classifier
            if fusion_cls:
                mod_classifiers['all-fused'] = nn.Sequential(nn.Dropout(dropout), nn.Linear(latent_dim, cls_dim)
                                                             ) if not common_classifier else common_classifier
            classifiers.update({cls_type: mod_classifiers})
        return classifiers

    @staticmethod
    def ordered_feature_list(x_d: Dict[str, Tensor], feats_order: List) -> List[Tensor]:
        return [x_d[modk] for modk in feats_order]

    def feature_mapping(self, x_d: Dict[str, Tensor]) -> Dict[str, Tensor]:
        return {modk: self.mapping[modk](x) for modk, x in x_d.items()}

    def _classify(self, head: nn.Sequential, feat: Tensor) -> Tensor:
        """Sequential(Dropout(p), Linear(d, classes)) on (B, T', d): dropout is applied while the features are
        staged as the GEMM operand, the Linear is the MFMA GEMM (N = classes, tail-masked)."""
        drop, lin = head[0], head[1]
        B, Tn, C = feat.shape
        in_drop = D_.elementwise(drop.p) if (self.training and drop.p > 0) else None
        y = F_.Linear.apply(feat.reshape(B * Tn, C), lin.weight, lin.bias, in_drop)
        return y.view(B, Tn, -1)

    def apply_classifier(self, input_feat, outputs_prefix=''):
        out = {}
        for classk in self.num_classes.keys():
            if classk in self.classifiers:
                out[f'{outputs_prefix}logits/{classk}'] = {
                    modk: self._classify(self.classifiers[classk][modk], input_feat[modk])
                    for modk in self.classifiers[classk].keys() if modk in input_feat}
            else:
                raise ValueError(f'Classifier for {classk} does not exist.')
        return out

    @staticmethod
    def prepare_output(z, z_hat, fusions):
        """orig_past / future / all-fused / past_futures bookkeeping (models/future_prediction.py:155-182)."""
        out = {'orig_past': z, 'future': z_hat, 'all-fused': fusions, 'past_futures': {}}
        B, T, C = next(iter(z.values())).shape
        for modk in out['future'].keys():
            out['past_futures'][modk] = torch.cat([out['orig_past'][modk][:, :1],
                                                   out['future'][modk][:, :(T - 1)]], dim=1)
            out['future'][modk] = out['future'][modk][:, (T - 1):]
        for modk in out['all-fused'].keys():
            out['all-fused'][modk] = out['all-fused'][modk][:, (T - 1):]
        return out

    @abc.abstractmethod
    def forward(self, x):
        raise NotImplementedError


class CMFPEarly(CrossModalFusionPrediction):
    """cross modality future predictor, early fusion version:
    features of different modalities are fused before the future prediction module"""

    def __init__(self, model_cfg, num_classes):
        logger = logging.getLogger(__name__)
        if not model_cfg.common.share_classifiers:
            logger.warning("Enforcing shared classifier for early CMFP.")
            model_cfg.common.share_classifiers = True
        if not model_cfg.common.share_predictors:
            logger.warning("Enforcing shared predictor for early CMFP.")
            model_cfg.common.share_predictors = True
        super().__init__(model_cfg, num_classes=num_classes)

    @staticmethod
    def _init_dimension_encoder(modality_dims, inter_dim, latent_dim):
        del modality_dims
        return nn.Linear(latent_dim, inter_dim, bias=False) if latent_dim != inter_dim else nn.Identity()

    @staticmethod
    def _init_dimension_decoder(modality_dims, inter_dim, latent_dim):
        del modality_dims
        return nn.Linear(inter_dim, latent_dim, bias=False) if latent_dim != inter_dim else nn.Identity()

    def forward(self, feats: Dict[str, torch.Tensor]) -> Dict[str, Dict[str, torch.Tensor]]:
        feats_order = [mod for mod in self.cfg.modal_feature_order if mod in feats]
        x_hat = self.feature_mapping(feats)
        order_feature_func = partial(self.ordered_feature_list, feats_order=feats_order)
        z, modality_attns = self.fuser(x_hat, order_feature_func)

        z_enc = self._project(self.dim_encoder, z)
        z_hat_enc, temporal_attns = self.future_predictor(z_enc, self.cfg.common.fp_output_len)
        z_hat = self._project(self.dim_decoder, z_hat_enc)

        z = {"all-fused": z}
        z_hat = {"all-fused": z_hat}
        attentions = {"all-fused": {'modality_attns': modality_attns, 'temporal_attns': temporal_attns}}
        fusion = {k: v[:] for k, v in z.items()}
        out = self.prepare_output(z, z_hat, fusion)
        feats_final = out["future"]
        out.update(self.apply_classifier(out["past_futures"], outputs_prefix=PAST_LOGITS_PREFIX))
        out.update(self.apply_classifier(feats_final))
        out['attentions'] = attentions
        return out


class IndividualFuturePrediction(CrossModalFusionPrediction):
    """Individual modality future predictor (models/future_prediction.py:189-225): no mapping, no fuser; every modality
    runs the (common or its own) GPT-2 predictor in its own width and is classified by its modality classifier."""

    def __init__(self, model_cfg, num_classes):
        assert not model_cfg.common.fusion_cls   # individual forwarding, fusion not possible
        super().__init__(model_cfg, num_classes=num_classes, instantiate_=False)
        self.future_predictor = self._init_future_predictor(model_cfg, self.common_predictor)

    def forward(self, z: Dict[str, torch.Tensor]) -> Dict[str, Dict[str, torch.Tensor]]:
        z = {k: (v if v.dtype == torch.float32 else v.float()) for k, v in z.items()}
        z_hat, _ = self._predict_unimodal(z)
        out = self.prepare_output(z, z_hat, {})   # in this case no fusion results
        feats_final = out["future"]
        out.update(self.apply_classifier(out["past_futures"], outputs_prefix=PAST_LOGITS_PREFIX))
        out.update(self.apply_classifier(feats_final))
        return out


class CMFPScoreFusion(CrossModalFusionPrediction):
    """Late fusion (models/future_prediction.py:294-351): per-modality prediction and classification, the class scores
    are mixed with the modality weights of the fuser (MATT) -- one fused weighted-sum kernel per logits tensor."""

    def __init__(self, model_cfg, num_classes):
        logger = logging.getLogger(__name__)
        assert not model_cfg.common.fusion_cls   # the classification scores are fused directly
        if not model_cfg.common.modality_cls:
            logger.warning("Enforcing modality classification for CMFPScoreFusion.")
            model_cfg.common.modality_cls = True
        super().__init__(model_cfg, num_classes=num_classes)

    def forward(self, z: Dict[str, torch.Tensor]) -> Dict[str, Dict[str, torch.Tensor]]:
        feats_order = [mod for mod in self.cfg.modal_feature_order if mod in z]
        z = {k: (v if v.dtype == torch.float32 else v.float()) for k, v in z.items()}
        z_hat, _ = self._predict_unimodal(z)
        # the first frame concatenated with the predicted frames, mapped to the common dim, gives the modality weights
        z_hat_cat = self.feature_mapping({modk: torch.cat([z[modk][:, :1, :], z_hat[modk]], dim=1) for modk in z})
        order_feature_func = partial(self.ordered_feature_list, feats_order=feats_order)
        modality_attns = self.fuser(z_hat_cat, order_feature_func)           # (B, T', M)
        out = self.prepare_output(z, z_hat, fusions={})
        logits_past = self.apply_classifier(out["past_futures"], outputs_prefix=PAST_LOGITS_PREFIX)
        logits_future = self.apply_classifier(out['future'])
        M = len(feats_order)
        w_past = modality_attns[:, :-1, :].reshape(-1, M)
        w_future = modality_attns[:, -1:, :].reshape(-1, M)
        for classk in self.num_classes.keys():
            lp = logits_past[f'{PAST_LOGITS_PREFIX}logits/{classk}']
            lf = logits_future[f'logits/{classk}']
            B, Tp, C = lp[feats_order[0]].shape
            past = F_.WeightedSum.apply(w_past, *[lp[m].reshape(B * Tp, C) for m in feats_order]).view(B, Tp, C)
            Tf = lf[feats_order[0]].shape[1]
            fut = F_.WeightedSum.apply(w_future, *[lf[m].reshape(B * Tf, C) for m in feats_order]).view(B, Tf, C)
            out[f'{PAST_LOGITS_PREFIX}logits/{classk}'] = {'all-fused': past}
            out[f'logits/{classk}'] = {'all-fused': fut}
        return out

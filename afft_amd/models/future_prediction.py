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
from .. import runtime as rt
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
        self.wpe.weight._afft_fp32_table = True      # read as fp32 rows (AddRowTable), never a GEMM image: parallel.FlatParams.owns_image
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
        return (outs[0] if len(outs) == 1 else torch.cat(outs, dim=1)), addl_endpoints


# --------------------------------------------------------------------------- cross-modal fusion + prediction
# The attribute names below (mapping, fuser, dim_encoder, dim_decoder, future_predictor, classifiers[cls][modality]) and the
# order in which they are registered are the reference's state_dict contract (models/future_prediction.py:19-186); how they
# are built and how the outputs are assembled is this file's own organisation: small builders instead of overridable
# _init_* hooks, one `_heads()` pass for every classifier application, one `_split_past_future()` for the bookkeeping.
def _resize(n_in: int, n_out: int) -> nn.Module:
    """bias-free width change of the GPT-2 side (dim_encoder / dim_decoder); nothing to do when the widths agree"""
    return nn.Identity() if n_in == n_out else nn.Linear(n_in, n_out, bias=False)


def _head(p_drop: float, n_in: int, n_classes: int) -> nn.Sequential:
    return nn.Sequential(nn.Dropout(p_drop), nn.Linear(n_in, n_classes))


def _as_rows_gemm(lin: nn.Module, x: Tensor) -> Tensor:
    """a `_resize` module on (B, T, C) through the MFMA GEMM"""
    if isinstance(lin, nn.Identity):
        return x
    B, T, C = x.shape
    return F_.Linear.apply(x.reshape(B * T, C), lin.weight, None).view(B, T, -1)


def _f32(t: Tensor) -> Tensor:
    return t if t.dtype == torch.float32 else t.float()


class CrossModalFusionPrediction(nn.Module, metaclass=abc.ABCMeta):
    """base class cross modality future predictor"""

    # late-fusion variants keep one encoder / decoder pair (and, unless shared, one predictor) PER MODALITY in that
    # modality's own width; the early-fusion variant has a single pair for the fused feature
    per_modality_codec = True

    def __init__(self, model_cfg, num_classes, instantiate_: bool = True, with_mapping_and_fuser: bool = True):
        super().__init__()
        assert is_dict_config(model_cfg.modal_dims), 'cfg.model.modal_dims must be a Dict!'
        common = model_cfg.common
        self.cfg, self.num_classes = model_cfg, num_classes
        self.latent_dim, self.fp_inter_dim = common.in_features, common.fp_inter_dim
        self.modality_dims = model_cfg.modal_dims
        self.common_predictor, self.common_classifier = common.share_predictors, common.share_classifiers
        self.modality_cls, self.fusion_cls = common.modality_cls, common.fusion_cls
        if instantiate_ and with_mapping_and_fuser:
            self.mapping = nn.ModuleDict()
            for mod, width in self.modality_dims.items():
                self.mapping[mod] = instantiate(model_cfg.mapping, in_features=width, out_features=self.latent_dim)
                logging.info(f'Using {self.mapping[mod]} for {mod}')
            self.fuser = instantiate(model_cfg.fuser, _recursive_=False)
        if instantiate_:
            self._build_predictor()
        self._build_classifiers()

    # ---- construction
    def _build_predictor(self):
        D, cfg = self.fp_inter_dim, self.cfg
        if self.per_modality_codec:
            self.dim_encoder = nn.ModuleDict({m: _resize(w, D) for m, w in self.modality_dims.items()})
            self.dim_decoder = nn.ModuleDict({m: _resize(D, w) for m, w in self.modality_dims.items()})
        else:
            self.dim_encoder, self.dim_decoder = _resize(self.latent_dim, D), _resize(D, self.latent_dim)
        make = lambda: instantiate(cfg.future_predictor, in_features=D, dimension_mapping=False, _recursive_=False)   # noqa: E731
        self.future_predictor = make() if self.common_predictor else nn.ModuleDict({m: make() for m in cfg.modal_dims.keys()})

    def _build_classifiers(self):
        assert self.modality_cls or self.fusion_cls, 'Modality-level and / or fusion classification!'
        p = self.cfg.dropout
        self.classifiers = nn.ModuleDict()
        for cls_type, n_cls in self.num_classes.items():
            shared = _head(p, self.latent_dim, n_cls) if self.common_classifier else None
            heads = nn.ModuleDict()
            if self.modality_cls:
                for m, width in self.modality_dims.items():
                    heads[m] = shared if shared is not None else _head(p, width, n_cls)
            if self.fusion_cls:
                heads['all-fused'] = shared if shared is not None else _head(p, self.latent_dim, n_cls)
            self.classifiers[cls_type] = heads

    # ---- pieces of forward
    @staticmethod
    def ordered_feature_list(x_d: Dict[str, Tensor], feats_order: List) -> List[Tensor]:
        return [x_d[modk] for modk in feats_order]

    def _modal_order(self, present) -> List[str]:
        return [m for m in self.cfg.modal_feature_order if m in present]

    def feature_mapping(self, x_d: Dict[str, Tensor]) -> Dict[str, Tensor]:
        return {modk: self.mapping[modk](x) for modk, x in x_d.items()}

    def _predict_unimodal(self, z: Dict[str, Tensor]):
        """every modality through (its) GPT-2 predictor in the predictor's width (models/future_prediction.py:204-217, :314-327)"""
        z_hat, attentions = {}, {}
        for m, feat in z.items():
            predictor = self.future_predictor if self.common_predictor else self.future_predictor[m]
            h, attentions[m] = predictor(_as_rows_gemm(self.dim_encoder[m], _f32(feat)), self.cfg.common.fp_output_len)
            z_hat[m] = _as_rows_gemm(self.dim_decoder[m], h)
        return z_hat, attentions

    def _classify(self, head: nn.Sequential, feat: Tensor) -> Tensor:
        """Sequential(Dropout(p), Linear(d, classes)) on (B, T', d): dropout is applied while the features are
        staged as the GEMM operand, the Linear is the MFMA GEMM (N = classes, tail-masked)."""
        drop, lin = head[0], head[1]
        B, Tn, C = feat.shape
        in_drop = D_.elementwise(drop.p) if (self.training and drop.p > 0) else None
        return F_.Linear.apply(feat.reshape(B * Tn, C), lin.weight, lin.bias, in_drop).view(B, Tn, -1)

    def apply_classifier(self, input_feat, outputs_prefix=''):
        """{prefix}logits/{class}: {modality: logits} for every classifier head that has an input (future_prediction.py:144-153)"""
        out = {}
        for classk in self.num_classes:
            if classk not in self.classifiers:
                raise ValueError(f'Classifier for {classk} does not exist.')
            heads = self.classifiers[classk]
            out[f'{outputs_prefix}logits/{classk}'] = {m: self._classify(heads[m], input_feat[m]) for m in heads if m in input_feat}
        return out

    @staticmethod
    def prepare_output(z, z_hat, fusions):
        """With T observed frames and predictions z_hat for frames 2 .. T + k (future_prediction.py:155-182):
        orig_past = z; past_futures = [z_1, z_hat_2 .. z_hat_T] (the observed first frame, then what was predicted for the
        other observed ones); future = z_hat_{T+1}.. ; all-fused = the fused feature from frame T on."""
        T = next(iter(z.values())).shape[1]
        cut = T - 1
        # [z_1, z_hat_2 .. z_hat_{T+k}] ONCE per modality: past_futures and future are its two ends (views) -- and the merged
        # classifier heads (_with_logits) take it whole, instead of a second concatenation of the same rows
        joined = {m: F_.SeenThenPredicted.apply(z[m], zh, T) for m, zh in z_hat.items()}
        out = {'orig_past': z,
               'future': {m: j[2] for m, j in joined.items()},
               'all-fused': {m: f[:, cut:] for m, f in fusions.items()},
               'past_futures': {m: j[1] for m, j in joined.items()}}
        out['_seen_then_predicted'] = {m: j[0] for m, j in joined.items()}
        return out

    def _with_logits(self, out: dict) -> dict:
        """past_logits/* from past_futures and logits/* from future (future_prediction.py:283-285 applies the SAME heads to both):
        one GEMM per head over the rows of both -- the future is 1-3 frames per clip, a GEMM of its own would be all launch and
        tail -- and two views of its output."""
        past, fut = out['past_futures'], out['future']
        whole = out.pop('_seen_then_predicted', None)
        if past.keys() == fut.keys():
            T = next(iter(past.values())).shape[1]
            if whole is None:
                whole = {m: torch.cat([past[m], fut[m]], dim=1) for m in past}
            both = self.apply_classifier(whole)
            for key, per_mod in both.items():
                halves = {m: F_.split_rows(v, T) for m, v in per_mod.items()}
                out[PAST_LOGITS_PREFIX + key] = {m: h[0] for m, h in halves.items()}
                out[key] = {m: h[1] for m, h in halves.items()}
            return out
        out.update(self.apply_classifier(past, outputs_prefix=PAST_LOGITS_PREFIX))
        out.update(self.apply_classifier(fut))
        return out

    @abc.abstractmethod
    def forward(self, x):
        raise NotImplementedError


class CMFPEarly(CrossModalFusionPrediction):
    """cross modality future predictor, early fusion version:
    features of different modalities are fused before the future prediction module"""
    per_modality_codec = False

    def __init__(self, model_cfg, num_classes):
        log = logging.getLogger(__name__)
        for key, what in (('share_classifiers', 'classifier'), ('share_predictors', 'predictor')):
            if not model_cfg.common[key]:
                log.warning(f"Enforcing shared {what} for early CMFP.")
                model_cfg.common[key] = True
        super().__init__(model_cfg, num_classes=num_classes)

    def forward(self, feats: Dict[str, torch.Tensor]) -> Dict[str, Dict[str, torch.Tensor]]:
        order = self._modal_order(feats)
        z, modality_attns = self.fuser(self.feature_mapping(feats), partial(self.ordered_feature_list, feats_order=order))
        h, temporal_attns = self.future_predictor(_as_rows_gemm(self.dim_encoder, z), self.cfg.common.fp_output_len)
        z_hat = _as_rows_gemm(self.dim_decoder, h)
        out = self._with_logits(self.prepare_output({'all-fused': z}, {'all-fused': z_hat}, {'all-fused': z}))
        out['attentions'] = {'all-fused': {'modality_attns': modality_attns, 'temporal_attns': temporal_attns}}
        return out


class IndividualFuturePrediction(CrossModalFusionPrediction):
    """Individual modality future predictor (models/future_prediction.py:189-225): no mapping, no fuser; every modality
    runs the (common or its own) GPT-2 predictor in its own width and is classified by its modality classifier."""

    def __init__(self, model_cfg, num_classes):
        assert not model_cfg.common.fusion_cls   # individual forwarding, fusion not possible
        super().__init__(model_cfg, num_classes=num_classes, with_mapping_and_fuser=False)

    def forward(self, z: Dict[str, torch.Tensor]) -> Dict[str, Dict[str, torch.Tensor]]:
        z = {k: _f32(v) for k, v in z.items()}
        z_hat, _ = self._predict_unimodal(z)
        return self._with_logits(self.prepare_output(z, z_hat, {}))   # no fusion results in this case


class CMFPScoreFusion(CrossModalFusionPrediction):
    """Late fusion (models/future_prediction.py:294-351): per-modality prediction and classification, the class scores
    are mixed with the modality weights of the fuser (MATT) -- one fused weighted-sum kernel per logits tensor."""

    def __init__(self, model_cfg, num_classes):
        assert not model_cfg.common.fusion_cls   # the classification scores are fused directly
        if not model_cfg.common.modality_cls:
            logging.getLogger(__name__).warning("Enforcing modality classification for CMFPScoreFusion.")
            model_cfg.common.modality_cls = True
        super().__init__(model_cfg, num_classes=num_classes)

    def forward(self, z: Dict[str, torch.Tensor]) -> Dict[str, Dict[str, torch.Tensor]]:
        order = self._modal_order(z)
        z = {k: _f32(v) for k, v in z.items()}
        z_hat, _ = self._predict_unimodal(z)
        # modality weights: the first observed frame followed by the predicted frames, mapped to the common width -> MATT
        seen_then_predicted = {m: torch.cat([z[m][:, :1, :], z_hat[m]], dim=1) for m in z}
        weights = self.fuser(self.feature_mapping(seen_then_predicted), partial(self.ordered_feature_list, feats_order=order))
        M = len(order)
        out = self._with_logits(self.prepare_output(z, z_hat, fusions={}))
        for classk in self.num_classes:
            for key, w in ((f'{PAST_LOGITS_PREFIX}logits/{classk}', weights[:, :-1, :]), (f'logits/{classk}', weights[:, -1:, :])):
                per_mod = out[key]
                B, Tn, C = per_mod[order[0]].shape
                mixed = F_.WeightedSum.apply(w.reshape(-1, M), *[per_mod[m].reshape(B * Tn, C) for m in order])
                out[key] = {'all-fused': mixed.view(B, Tn, C)}
        return out

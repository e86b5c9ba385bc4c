"""Builds the ``model_cfg`` tree the reference gets from Hydra (conf/model/* + expts/*.txt overrides, with
the ``${...}`` interpolations resolved) for use without Hydra: tests, smoke and bench."""
from __future__ import annotations

from typing import Dict, Optional

from ._hydra_compat import to_attr

MODAL_FEATURE_ORDER = ["rgb", "objects", "audio", "poses", "flow"]  # conf/config.yaml:41


def make_model_cfg(modal_dims: Dict[str, int], common_dim: int, fp_inter_dim: int = 2048, fuser: str = "sa",
                   depth: int = 6, num_heads: int = 4, fp_layers: int = 6, fp_heads: int = 4, fp_output_len: int = 1,
                   dropout: float = 0.2, drop: float = 0.1, cross_attn: bool = False, modal_encoding: bool = False,
                   frame_level_token: bool = False, T: Optional[int] = None, fp_output_attentions: bool = False,
                   cmfp: str = "early", mapping: str = "linear", mapping_activation: str = "relu",
                   mapping_layernorm: Optional[bool] = None, share_predictors: bool = True,
                   share_classifiers: bool = True):
    """conf/model/{common,fuser/SA-Fuser|CA-Fuser,future_predictor/base_future_predictor,CMFP/cmfp_early,
    mapping/linear}.yaml with expts/01 (SA) / expts/04 (CA) overrides. `drop` sets every transformer dropout
    and DropPath rate (reference value 0.1)."""
    if fuser == "sa":
        fz = dict(_target_="models.fusion.ModalTokenCMFuser", dim=common_dim, depth=depth, num_heads=num_heads,
                  embd_drop_rate=drop, drop_rate=drop, attn_drop_rate=drop, drop_path_rate=drop,
                  cross_attn=cross_attn, norm_elementwise=True, modalities=dict(modal_dims),
                  modal_encoding=modal_encoding, frame_level_token=frame_level_token,
                  temporal_sequence_length=T if frame_level_token else None)
    elif fuser == "ca":
        fz = dict(_target_="models.fusion.TemporalCrossAttentFuser", dim=common_dim, modalities=dict(modal_dims),
                  num_heads=num_heads, embd_drop_rate=drop, drop_rate=drop, attn_drop_rate=drop, drop_path_rate=drop)
    elif fuser == "cm":      # conf/model/fuser/CMFuser.yaml: SA-Fuser without modality token
        fz = dict(_target_="models.fusion.CMFuser", dim=common_dim, depth=depth, num_heads=num_heads,
                  embd_drop_rate=drop, drop_rate=drop, attn_drop_rate=drop, drop_path_rate=drop, cross_attn=cross_attn)
    elif fuser == "tsa":     # conf/model/fuser/T-SA-Fuser.yaml
        fz = dict(_target_="models.fusion.TemporalCMFuser", dim=common_dim, depth=depth, num_heads=num_heads,
                  embd_drop_rate=drop, drop_rate=drop, attn_drop_rate=drop, drop_path_rate=drop,
                  modalities=dict(modal_dims), modal_encoding=modal_encoding, frame_level_token=frame_level_token,
                  temporal_sequence_length=T if frame_level_token else None)
    elif fuser == "matt":    # conf/model/fuser/MATT.yaml (late score fusion)
        fz = dict(_target_="models.fusion.MATT", modal_dims=dict(modal_dims), dim=common_dim, drop_rate=0.8)
    elif fuser == "none":
        fz = dict(_target_="torch.nn.Identity")
    else:
        raise ValueError(fuser)
    late = cmfp != "early"
    mp = {"linear": dict(_target_="models.feature_mapping.Linear",
                         use_layernorm=bool(mapping_layernorm) if mapping_layernorm is not None else False,
                         sparse_mapping=True),
          "nonlinear": dict(_target_="models.feature_mapping.NonLinear",
                            use_layernorm=bool(mapping_layernorm) if mapping_layernorm is not None else False,
                            activation=mapping_activation),
          "gated": dict(_target_="models.feature_mapping.GatedLinear",
                        use_layernorm=bool(mapping_layernorm) if mapping_layernorm is not None else True)}[mapping]
    cmfp_target = {"early": "models.future_prediction.CMFPEarly", "score": "models.future_prediction.CMFPScoreFusion",
                   "individual": "models.future_prediction.IndividualFuturePrediction"}[cmfp]
    cfg = dict(
        modal_dims=dict(modal_dims), modal_feature_order=list(MODAL_FEATURE_ORDER), common_dim=common_dim,
        dropout=dropout,
        common=dict(in_features=common_dim, share_classifiers=share_classifiers, share_predictors=share_predictors,
                    modality_cls=late, fusion_cls=not late,
                    backbones={m: {"_target_": "torch.nn.Identity"} for m in modal_dims},
                    fp_output_len=fp_output_len, fp_inter_dim=fp_inter_dim, fp_layers=fp_layers, fp_heads=fp_heads,
                    fp_output_attentions=fp_output_attentions, embd_pdrop=drop, resid_pdrop=drop, attn_pdrop=drop),
        mapping=mp,
        fuser=fz,
        future_predictor=dict(_target_="models.future_prediction.BaseFuturePredictor", in_features=common_dim,
                              inter_dim=fp_inter_dim, n_layer=fp_layers, n_head=fp_heads,
                              output_attentions=fp_output_attentions, embd_pdrop=drop, resid_pdrop=drop,
                              attn_pdrop=drop),
        CMFP=dict(_target_=cmfp_target, model_cfg=None),
    )
    return to_attr(cfg)


# BASELINE.json configs: name -> kwargs for make_model_cfg + (B, T)
BASELINE_CONFIGS = {
    "cfg1": dict(modal_dims={"rgb": 1024, "flow": 1024}, common_dim=1024, fp_inter_dim=2048, fuser="sa", T=8),
    "ek100": dict(modal_dims={"rgb": 1024, "objects": 352, "audio": 1024, "flow": 1024}, common_dim=1024,
                  fp_inter_dim=2048, fuser="sa", T=16),
    "cfg2": dict(modal_dims={"rgb": 2048, "objects": 2048, "audio": 2048, "flow": 2048}, common_dim=2048,
                 fp_inter_dim=2048, fuser="sa", T=16),
    "cfg4": dict(modal_dims={"rgb": 2048, "objects": 2048, "audio": 2048, "flow": 2048}, common_dim=2048,
                 fp_inter_dim=2048, fuser="ca", T=16),
    "cfg5": dict(modal_dims={"rgb": 2048, "objects": 2048, "audio": 2048, "poses": 2048, "flow": 2048},
                 common_dim=2048, fp_inter_dim=2048, fuser="sa", T=32),
    # not BASELINE rows: the other fusers at the cfg2 sizes, for DESIGN.md section 7 (depth 6 like the SA-Fuser)
    "cfg2_cm": dict(modal_dims={"rgb": 2048, "objects": 2048, "audio": 2048, "flow": 2048}, common_dim=2048,
                    fp_inter_dim=2048, fuser="cm", T=16),
    "cfg2_tsa": dict(modal_dims={"rgb": 2048, "objects": 2048, "audio": 2048, "flow": 2048}, common_dim=2048,
                     fp_inter_dim=2048, fuser="tsa", T=16),
}


def gflop_per_clip(name_or_cfg, fwd_bwd: bool = True, executed: bool = False, token_row_projection: bool = False) -> float:
    """Algorithmic FLOPs per clip (2*MAC), SURVEY.md 8(d) formulas = what the reference executes.
    executed=True: what this implementation executes with runtime.skip_dead_rows() on -- the SA-Fuser's last block runs its MLP
    half on token 0 of every frame only (16 * T * (S - 1) * d^2 less; the other rows never reach an output) and, with
    token_row_projection (round 5: batches whose frame count is a multiple of 64 on the composite path, functional.attn_take_ok),
    its attention output projection too (another 2 * T * (S - 1) * d^2)."""
    c = BASELINE_CONFIGS[name_or_cfg] if isinstance(name_or_cfg, str) else name_or_cfg
    md, d, D, T = c["modal_dims"], c["common_dim"], c.get("fp_inter_dim", 2048), c["T"]
    M = len(md)
    depth, gl, ncls = c.get("depth", 6), c.get("fp_layers", 6), c.get("num_classes", 3806)
    fl = 0.0
    if c.get("fuser", "sa") == "sa":
        S = M + 1
        fl += depth * T * S * 24 * d * d + depth * T * 4 * S * S * d
        if executed and depth >= 1:
            fl -= T * (S - 1) * 16 * d * d
            if token_row_projection:
                fl -= T * (S - 1) * 2 * d * d
    elif c["fuser"] == "cm":       # M tokens per frame
        fl += depth * T * M * 24 * d * d + depth * T * 4 * M * M * d
    elif c["fuser"] == "tsa":      # one sequence of M*T tokens per clip
        fl += depth * T * M * 24 * d * d + depth * 4 * (M * T) * (M * T) * d
    else:
        fl += (M - 1) * T * 32 * d * d + (M - 1) * 8 * T * T * d
    fl += sum(2 * T * C * d for C in md.values() if C != d)
    if d != D:
        fl += 4 * T * d * D
    fl += gl * T * 24 * D * D + gl * 4 * T * T * D
    fl += 2 * (T + 1) * d * ncls
    return fl * (3.0 if fwd_bwd else 1.0) / 1e9

"""Golden-fixture case table shared by make_golden.py (reference side, build container only)
and the parity tests (oracle / HIP side, anywhere)."""

_MODS4 = {"rgb": 64, "objects": 24, "audio": 64, "flow": 64}

CASES = {
    # T0: tiny SA-Fuser, every reference width ratio kept (objects needs a mapping GEMM with K=24,
    # d != D so dim_encoder/decoder exist), hard labels with ignored (-1) past frames.
    "t0_sa": dict(fuser="sa", modal_dims=_MODS4, d=64, D=128, depth=2, num_heads=4, fp_layers=2, fp_heads=2,
                  T=4, B=3, num_classes=11, fp_output_len=1),
    # T1: tiny CA-Fuser (depth = M-1 = 3 DecoderBlocks), causal self + cross attention.
    "t1_ca": dict(fuser="ca", modal_dims=_MODS4, d=64, D=128, num_heads=4, fp_layers=2, fp_heads=2,
                  T=4, B=3, num_classes=11, fp_output_len=1),
    # T2 variants
    "t2_diag": dict(fuser="sa", modal_dims=_MODS4, d=64, D=128, depth=2, num_heads=4, fp_layers=2, fp_heads=2,
                    T=4, B=3, num_classes=11, fp_output_len=1, cross_attn=True),
    "t2_flt": dict(fuser="sa", modal_dims=_MODS4, d=64, D=128, depth=2, num_heads=4, fp_layers=2, fp_heads=2,
                   T=4, B=3, num_classes=11, fp_output_len=1, frame_level_token=True, modal_encoding=True),
    "t2_roll": dict(fuser="sa", modal_dims={"rgb": 64, "flow": 64}, d=64, D=64, depth=1, num_heads=2, fp_layers=2,
                    fp_heads=2, T=4, B=2, num_classes=7, fp_output_len=3),
    # soft targets (MixUp replayed with a fixed lambda) + ignore rows
    "t2_soft": dict(fuser="sa", modal_dims=_MODS4, d=64, D=128, depth=2, num_heads=4, fp_layers=2, fp_heads=2,
                    T=4, B=4, num_classes=11, fp_output_len=1, soft=True, lam=0.3, label_smoothing=0.4,
                    ignore_frac=0.1),
    # identity everywhere (all modal dims == d == D), M=5 with 'poses', T=8: the BASELINE cfg5 topology in small
    "t3_m5": dict(fuser="sa", modal_dims={"rgb": 64, "objects": 64, "audio": 64, "poses": 64, "flow": 64},
                  d=64, D=64, depth=2, num_heads=4, fp_layers=2, fp_heads=4, T=8, B=2, num_classes=13,
                  fp_output_len=1),
    # "next" fusers.  T4: CMFuser = SA-Fuser without modality token (mean over the M tokens), with the
    # -inf diagonal mask; T5: T-SA-Fuser, M*T = 16 / 20 tokens per clip under the causal mask tiled over the modalities,
    # with position + modality embeddings; once averaging the modality tokens, once with frame-level modal tokens.
    "t4_cm": dict(fuser="cm", modal_dims=_MODS4, d=64, D=128, depth=2, num_heads=4, fp_layers=2, fp_heads=2,
                  T=4, B=3, num_classes=11, fp_output_len=1, cross_attn=True),
    "t5_tsa": dict(fuser="tsa", modal_dims=_MODS4, d=64, D=128, depth=2, num_heads=4, fp_layers=2, fp_heads=2,
                   T=4, B=3, num_classes=11, fp_output_len=1, modal_encoding=True),
    "t5_tsa_tok": dict(fuser="tsa", modal_dims=_MODS4, d=64, D=128, depth=2, num_heads=4, fp_layers=2, fp_heads=2,
                       T=4, B=3, num_classes=11, fp_output_len=1, modal_encoding=True, frame_level_token=True),
    # a 40-token T-SA sequence (M = 4, T = 10 as in expts/): exercises the L > 32 attention kernels
    "t5_tsa_l40": dict(fuser="tsa", modal_dims={"rgb": 64, "objects": 64, "audio": 64, "flow": 64}, d=64, D=64,
                       depth=1, num_heads=2, fp_layers=1, fp_heads=2, T=10, B=2, num_classes=7, fp_output_len=1,
                       modal_encoding=True),
    # late fusion / per-modality heads and the other mapping layers.  Modality widths differ (24-wide
    # objects): per-modality dim_encoder / dim_decoder and classifiers; MATT weights mix the class scores.
    "t6_score": dict(cmfp="score", fuser="matt", modal_dims={"rgb": 64, "objects": 24, "flow": 64}, d=64, D=128,
                     num_heads=0, fp_layers=2, fp_heads=2, T=4, B=3, num_classes=11, fp_output_len=1,
                     share_predictors=True, share_classifiers=False),
    "t6_score_own": dict(cmfp="score", fuser="matt", modal_dims={"rgb": 64, "flow": 64}, d=64, D=64, num_heads=0,
                         fp_layers=1, fp_heads=2, T=4, B=2, num_classes=7, fp_output_len=1, share_predictors=False,
                         share_classifiers=False, mapping="nonlinear", mapping_activation="relu"),
    "t7_indiv": dict(cmfp="individual", fuser="none", modal_dims={"rgb": 64, "objects": 24}, d=64, D=128, num_heads=0,
                     fp_layers=2, fp_heads=2, T=4, B=3, num_classes=11, fp_output_len=1, share_predictors=True,
                     share_classifiers=False),
    "t8_gated": dict(fuser="sa", modal_dims=_MODS4, d=64, D=128, depth=1, num_heads=4, fp_layers=1, fp_heads=2,
                     T=4, B=3, num_classes=11, fp_output_len=1, mapping="gated", mapping_layernorm=True),
    "t8_nonlin": dict(fuser="sa", modal_dims=_MODS4, d=64, D=128, depth=1, num_heads=4, fp_layers=1, fp_heads=2,
                      T=4, B=3, num_classes=11, fp_output_len=1, mapping="nonlinear", mapping_activation="gelu",
                      mapping_layernorm=True),
}

# Full-size fixtures (SURVEY.md 8c "F"): the REFERENCE itself run at the real widths -- head dims 256 / 512, 3806 classes, 6 + 6
# layers, 388-614 M parameters -- stored compactly (small tensors whole; large ones as norm + sum + a strided sample; every
# parameter's gradient as norm + a 256-element strided sample).  B = 2 so that reference and oracle take seconds on a CPU.
_DEPTH66 = dict(depth=6, num_heads=4, fp_layers=6, fp_heads=4, num_classes=3806, fp_output_len=1)
FULL_CASES = {
    # BASELINE configs[0]: rgb + flow, T = 8, d = 1024, D = 2048, B = 4 (the reference's own CPU-runnable case)
    "f_cfg1": dict(fuser="sa", modal_dims={"rgb": 1024, "flow": 1024}, d=1024, D=2048, T=8, B=4, **_DEPTH66),
    # what expts/01_SA-Fuser_ek100_train.txt trains: 1024 / 352 / 1024 / 1024 -> d = 1024, D = 2048, T = 16
    "f_ek100": dict(fuser="sa", modal_dims={"rgb": 1024, "objects": 352, "audio": 1024, "flow": 1024}, d=1024, D=2048,
                    T=16, B=2, **_DEPTH66),
    # BASELINE configs[1] (the bench workload): every width 2048
    "f_cfg2": dict(fuser="sa", modal_dims={"rgb": 2048, "objects": 2048, "audio": 2048, "flow": 2048}, d=2048, D=2048,
                   T=16, B=2, **_DEPTH66),
    # BASELINE configs[3]: CA-Fuser (TemporalCrossAttentFuser, depth M - 1 = 3) at the bench widths
    "f_cfg4": dict(fuser="ca", modal_dims={"rgb": 2048, "objects": 2048, "audio": 2048, "flow": 2048}, d=2048, D=2048,
                   T=16, B=2, num_heads=4, fp_layers=6, fp_heads=4, num_classes=3806, fp_output_len=1),
}
# BASELINE configs[4]: five modalities (+ poses), T = 32 (two 16-row attention tiles per sequence in the predictor, S = 6 packing)
FULL_CASES["f_cfg5"] = dict(fuser="sa", modal_dims={"rgb": 2048, "objects": 2048, "audio": 2048, "poses": 2048, "flow": 2048},
                            d=2048, D=2048, T=32, B=2, **_DEPTH66)
FULL_MAX_WHOLE = 70000      # tensors up to this many elements are stored whole
FULL_SAMPLES = 16384        # larger ones: this many strided samples
FULL_GRAD_SAMPLES = 256

GRAD_KEYS_SA = [
    "future_predictor.fuser.modal_token",
    "future_predictor.fuser.blocks.0.attn.qkv.weight",
    "future_predictor.fuser.blocks.0.norm1.weight",
    "future_predictor.fuser.blocks.1.mlp.mlp.2.bias",
    "future_predictor.future_predictor.gpt_model.wpe.weight",
    "future_predictor.future_predictor.gpt_model.h.0.attn.c_attn.weight",
    "future_predictor.future_predictor.gpt_model.h.1.mlp.c_fc.bias",
    "future_predictor.future_predictor.gpt_model.ln_f.weight",
    "future_predictor.classifiers.action.all-fused.1.weight",
]
GRAD_KEYS_CA = [
    "future_predictor.fuser.position_embeddings.weight",
    "future_predictor.fuser.blocks.0.cross_attn.w_k.weight",
    "future_predictor.fuser.blocks.1.attn.qkv.weight",
    "future_predictor.fuser.blocks.2.norm_kv.bias",
    "future_predictor.fuser.blocks.2.mlp.mlp.0.weight",
    "future_predictor.future_predictor.gpt_model.h.0.attn.c_proj.weight",
    "future_predictor.classifiers.action.all-fused.1.bias",
]
GRAD_KEYS_CM = [k for k in GRAD_KEYS_SA if "modal_token" not in k] + ["future_predictor.fuser.norm.weight"]
GRAD_KEYS_TSA = GRAD_KEYS_CM + ["future_predictor.fuser.position_embeddings.weight"]
GRAD_KEYS_LATE = [
    "future_predictor.dim_encoder.objects.weight",
    "future_predictor.dim_decoder.objects.weight",
    "future_predictor.dim_encoder.rgb.weight",
    "future_predictor.future_predictor.gpt_model.h.0.attn.c_attn.weight",
    "future_predictor.future_predictor.rgb.gpt_model.h.0.mlp.c_fc.weight",
    "future_predictor.future_predictor.flow.gpt_model.wpe.weight",
    "future_predictor.classifiers.action.rgb.1.weight",
    "future_predictor.classifiers.action.objects.1.weight",
    "future_predictor.classifiers.action.flow.1.bias",
    "future_predictor.fuser.matt.0.weight",
    "future_predictor.fuser.matt.3.bias",
    "future_predictor.fuser.matt.6.weight",
    "future_predictor.mapping.objects.mapping.0.weight",
    "future_predictor.mapping.rgb.mapping.0.bias",
]
OPTIONAL_GRAD_KEYS = [
    "future_predictor.mapping.rgb.mapping.0.weight",
    "future_predictor.mapping.rgb.mapping.0.bias",
    "future_predictor.mapping.objects.mapping.1.fc.weight",
    "future_predictor.mapping.objects.mapping.1.fc.bias",
    "future_predictor.mapping.flow.mapping.2.weight",
    "future_predictor.mapping.audio.mapping.2.bias",
    "future_predictor.fuser.modal_token",
    "future_predictor.mapping.objects.mapping.0.weight",
    "future_predictor.dim_encoder.weight",
    "future_predictor.dim_decoder.weight",
    "future_predictor.fuser.modality_embedding",
]


def grad_keys(c: dict):
    if c.get("cmfp", "early") != "early":
        return GRAD_KEYS_LATE
    return {"sa": GRAD_KEYS_SA, "ca": GRAD_KEYS_CA, "cm": GRAD_KEYS_CM, "tsa": GRAD_KEYS_TSA}[c["fuser"]] + OPTIONAL_GRAD_KEYS


def oracle_cfg(c: dict) -> dict:
    return dict(fuser=c["fuser"], depth=c.get("depth", 0), num_heads=c["num_heads"], fp_layers=c["fp_layers"],
                fp_heads=c["fp_heads"], fp_output_len=c.get("fp_output_len", 1),
                cross_attn=c.get("cross_attn", False), frame_level_token=c.get("frame_level_token", False),
                num_classes={"action": c["num_classes"]}, cmfp=c.get("cmfp", "early"),
                share_predictors=c.get("share_predictors", True), mapping=c.get("mapping", "linear"),
                mapping_activation=c.get("mapping_activation", "relu"))

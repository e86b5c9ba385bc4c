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
}

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
OPTIONAL_GRAD_KEYS = [
    "future_predictor.fuser.modal_token",
    "future_predictor.mapping.objects.mapping.0.weight",
    "future_predictor.dim_encoder.weight",
    "future_predictor.dim_decoder.weight",
    "future_predictor.fuser.modality_embedding",
]


def oracle_cfg(c: dict) -> dict:
    return dict(fuser=c["fuser"], depth=c.get("depth", 0), num_heads=c["num_heads"], fp_layers=c["fp_layers"],
                fp_heads=c["fp_heads"], fp_output_len=c.get("fp_output_len", 1),
                cross_attn=c.get("cross_attn", False), frame_level_token=c.get("frame_level_token", False),
                num_classes={"action": c["num_classes"]})

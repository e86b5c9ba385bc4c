"""Generate golden vectors by running the REFERENCE itself (build container only).

    python tests/golden/make_golden.py            # writes tests/golden/<case>.npz

Imports /root/reference's models/* and common/runner.py on CPU with thin stubs for the
packages the image lacks (hydra, omegaconf, timm, submitit, cv2 -- SURVEY.md 8c / Appendix B),
loads closed-form weights (closed_form.py), runs BaseModel.forward + BasicLossAccuracy +
backward in eval mode (all dropouts inactive), cross-checks this repo's oracle against the
reference to <=2e-5, and stores the reference's outputs.  Nothing of the reference (source,
bytecode, pickles) is written: the .npz files hold numeric arrays only.

Versions used for the committed fixtures are recorded inside each .npz ('meta').
"""
from __future__ import annotations

import importlib
import json
import os
import sys
import types

import numpy as np
import torch
import transformers  # must be imported BEFORE the timm stub (HF probes timm via find_spec)

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = os.environ.get("AFFT_REFERENCE", "/root/reference")


# ----------------------------------------------------------------------------- stubs
def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class DictConfig(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    __setattr__ = dict.__setitem__


def _wrap(o):
    if isinstance(o, dict):
        return DictConfig({k: _wrap(v) for k, v in o.items()})
    return o


def instantiate(cfg, *args, **kw):
    kw.pop("_recursive_", None)
    cfg = dict(cfg)
    modname, cls = cfg.pop("_target_").rsplit(".", 1)
    cfg.update(kw)
    return getattr(importlib.import_module(modname), cls)(*args, **cfg)


def install_stubs():
    _mod("timm")
    _mod("timm.models")
    _mod("timm.models.layers", trunc_normal_=torch.nn.init.trunc_normal_)
    _mod("omegaconf", OmegaConf=type("OmegaConf", (), {}), DictConfig=DictConfig, ListConfig=list)
    hu = _mod("hydra.utils", instantiate=instantiate, call=instantiate)
    _mod("hydra", utils=hu, main=lambda **k: (lambda f: f))
    _mod("hydra.types", TargetConf=dict)

    class _JE:
        def __init__(self):
            raise RuntimeError("no submitit")

    _mod("submitit", JobEnvironment=_JE)
    _mod("cv2")
    sys.path.insert(0, REF)


class CudaToCpu(torch.overrides.TorchFunctionMode):
    """The reference hard-codes device='cuda' in some fusers (models/fusion.py:254-255,332)."""

    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        if kwargs.get("device") in ("cuda", torch.device("cuda")):
            kwargs["device"] = "cpu"
        args = tuple("cpu" if (isinstance(a, str) and a == "cuda") else a for a in args)
        return func(*args, **kwargs)


# ----------------------------------------------------------------------------- model cfg
def build_model_cfg(c: dict) -> DictConfig:
    mods = c["modal_dims"]
    if c["fuser"] == "sa":
        fuser = dict(_target_="models.fusion.ModalTokenCMFuser", dim=c["d"], depth=c["depth"],
                     num_heads=c["num_heads"], embd_drop_rate=0.1, drop_rate=0.1, attn_drop_rate=0.1,
                     drop_path_rate=0.1, cross_attn=c.get("cross_attn", False), norm_elementwise=True,
                     modalities=_wrap(dict(mods)), modal_encoding=c.get("modal_encoding", False),
                     frame_level_token=c.get("frame_level_token", False),
                     temporal_sequence_length=c["T"] if c.get("frame_level_token") else None)
    elif c["fuser"] == "cm":
        fuser = dict(_target_="models.fusion.CMFuser", dim=c["d"], depth=c["depth"], num_heads=c["num_heads"],
                     embd_drop_rate=0.1, drop_rate=0.1, attn_drop_rate=0.1, drop_path_rate=0.1,
                     cross_attn=c.get("cross_attn", False))
    elif c["fuser"] == "tsa":
        fuser = dict(_target_="models.fusion.TemporalCMFuser", dim=c["d"], depth=c["depth"], num_heads=c["num_heads"],
                     embd_drop_rate=0.1, drop_rate=0.1, attn_drop_rate=0.1, drop_path_rate=0.1,
                     modalities=_wrap(dict(mods)), modal_encoding=c.get("modal_encoding", True),
                     frame_level_token=c.get("frame_level_token", False),
                     temporal_sequence_length=c["T"] if c.get("frame_level_token") else None)
    elif c["fuser"] == "matt":
        fuser = dict(_target_="models.fusion.MATT", modal_dims=_wrap(dict(mods)), dim=c["d"], drop_rate=0.8)
    elif c["fuser"] == "none":
        fuser = dict(_target_="torch.nn.Identity")
    else:
        fuser = dict(_target_="models.fusion.TemporalCrossAttentFuser", dim=c["d"], modalities=_wrap(dict(mods)),
                     num_heads=c["num_heads"], embd_drop_rate=0.1, drop_rate=0.1, attn_drop_rate=0.1,
                     drop_path_rate=0.1)
    late = c.get("cmfp", "early") != "early"
    mapping = {"linear": dict(_target_="models.feature_mapping.Linear", use_layernorm=c.get("mapping_layernorm", False),
                              sparse_mapping=True),
               "nonlinear": dict(_target_="models.feature_mapping.NonLinear", use_layernorm=c.get("mapping_layernorm", False),
                                 activation=c.get("mapping_activation", "relu")),
               "gated": dict(_target_="models.feature_mapping.GatedLinear",
                             use_layernorm=c.get("mapping_layernorm", True))}[c.get("mapping", "linear")]
    cmfp_target = {"early": "models.future_prediction.CMFPEarly", "score": "models.future_prediction.CMFPScoreFusion",
                   "individual": "models.future_prediction.IndividualFuturePrediction"}[c.get("cmfp", "early")]
    cfg = dict(
        modal_dims=dict(mods),
        modal_feature_order=["rgb", "objects", "audio", "poses", "flow"],
        common_dim=c["d"], dropout=0.2,
        common=dict(in_features=c["d"], share_classifiers=c.get("share_classifiers", True),
                    share_predictors=c.get("share_predictors", True), modality_cls=late,
                    fusion_cls=not late, backbones={m: {"_target_": "torch.nn.Identity"} for m in mods},
                    fp_output_len=c.get("fp_output_len", 1), fp_inter_dim=c["D"], fp_layers=c["fp_layers"],
                    fp_heads=c["fp_heads"], fp_output_attentions=False, embd_pdrop=0.1, resid_pdrop=0.1,
                    attn_pdrop=0.1),
        mapping=mapping,
        fuser=fuser,
        future_predictor=dict(_target_="models.future_prediction.BaseFuturePredictor", in_features=c["d"],
                              inter_dim=c["D"], n_layer=c["fp_layers"], n_head=c["fp_heads"],
                              output_attentions=False, embd_pdrop=0.1, resid_pdrop=0.1, attn_pdrop=0.1),
        CMFP=dict(_target_=cmfp_target, model_cfg=None),
    )
    return _wrap(cfg)


def flatten_outputs(out: dict) -> dict:
    flat = {}
    for k, v in out.items():
        if k == "attentions":
            ma = v["all-fused"]["modality_attns"]
            flat["attentions/modality_attns"] = ma
            continue
        for kk, t in v.items():
            flat[f"{k}/{kk}"] = t
    return flat


def surrogate(out: dict):
    """Scalar used for gradient parity where the reference's loss is undefined (fp_output_len > 1)."""
    return (out["logits/action"]["all-fused"].pow(2).mean() + out["past_logits/action"]["all-fused"].pow(2).mean()
            + out["past_futures"]["all-fused"].pow(2).mean())


def run_case(name: str, c: dict):
    import closed_form as cf
    from cases import grad_keys, oracle_cfg
    from models.base_model import BaseModel
    from common.runner import BasicLossAccuracy, Runner
    from oracle import afft_oracle as O

    torch.manual_seed(0)
    with CudaToCpu():
        model = BaseModel(build_model_cfg(c), num_classes={"action": c["num_classes"]}, class_mappings={})
    model.eval()
    sd = model.state_dict()
    shapes = {k: tuple(v.shape) for k, v in sd.items() if v.dtype.is_floating_point and not k.endswith(".attn.bias")
              and not k.endswith("masked_bias")}
    state = cf.fill_state(shapes)
    missing = model.load_state_dict(state, strict=False)
    assert not missing.unexpected_keys, missing

    B, T, K = c["B"], c["T"], c["num_classes"]
    data = cf.inputs_for(name, c["modal_dims"], B, T)
    tgt, sub = cf.labels_for(name, B, T, K, c.get("ignore_frac", 0.25))

    kwargs = dict(mixup_fn=None, target={"action": tgt}, target_subclips={"action": sub},
                  target_subclips_ignore_index=None)
    soft = c.get("soft", False)
    lam = c.get("lam")
    if soft:
        # Replay MixUp deterministically: the reference's MixUp with its Beta sample pinned to lam.
        from common.mixup import MixUp
        mix = MixUp(alpha=0.1, label_smoothing={"action": c["label_smoothing"]}, num_classes={"action": K})

        class _Fixed:
            def sample(self_inner):
                return torch.tensor(lam)

        mix.mixup_beta_sampler = _Fixed()
        kwargs["mixup_fn"] = mix

    roll = c.get("fp_output_len", 1) > 1
    with CudaToCpu():
        outputs, out_t = model({m: d.clone() for m, d in data.items()}, **kwargs)
        if not roll:
            losses, _ = BasicLossAccuracy()(outputs, out_t["target"], out_t["target_subclips"], mixup_enable=soft,
                                             target_subclips_ignore_index=out_t["target_subclips_ignore_index"])
    wts = {"cls_action": 1.0, "past_cls_action": 1.0, "past_reg": 1.0}
    if roll:
        # the reference's loss needs dense (B,T') targets when fp_output_len > 1 (common/runner.py:57);
        # its configs never use that, so the roll-out case pins outputs and a surrogate-scalar gradient only.
        losses = {}
        total = surrogate(outputs)
    else:
        total, lm = Runner._reduce_loss(losses, wts)
    model.zero_grad()
    total.backward()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}

    # ---- cross-check our oracle against the reference on the same weights / inputs
    P = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    ocfg = oracle_cfg(c)
    feats_in = {m: d.clone() for m, d in data.items()}
    if soft:
        f3 = {m: torch.flatten(d.mean([-1, -2]).permute(0, 1, 3, 2), 1, 2) for m, d in feats_in.items()}
        f3, t_soft, s_soft, ign = O.mixup(f3, tgt, sub, K, c["label_smoothing"], lam)
        oout = O.cmfp_early(P, f3, ocfg)
        ototal, olosses = O.loss(oout, t_soft, s_soft, soft=True, ignore=ign)
    elif roll:
        oout = O.base_model_forward(P, feats_in, ocfg)
        ototal = surrogate(oout)
    else:
        oout = O.base_model_forward(P, feats_in, ocfg)
        ototal, olosses = O.loss(oout, tgt, sub)
    ototal.backward()
    ref_flat = flatten_outputs(outputs)
    ora_flat = flatten_outputs(oout)
    worst = 0.0
    for k, v in ref_flat.items():
        if k == "attentions/modality_attns" and c["fuser"] == "ca":
            continue
        e = (ora_flat[k].detach() - v.detach()).norm() / (v.detach().norm() + 1e-30)
        worst = max(worst, float(e))
        assert e < 2e-5, (name, k, float(e))
    assert abs(float(ototal) - float(total)) < 2e-5 * max(1.0, abs(float(total))), (float(ototal), float(total))
    for k, g in grads.items():
        og = P[k].grad
        e = (og - g).norm() / (g.norm() + 1e-30)
        worst = max(worst, float(e))
        assert e < 5e-5, (name, "grad", k, float(e))
    print(f"[{name}] oracle == reference: worst rel-L2 {worst:.2e}; loss {float(total):.6f}")

    # ---- store reference outputs
    keys = grad_keys(c)
    arrays = {}
    for k, v in ref_flat.items():
        arrays["out:" + k] = v.detach().numpy().astype(np.float32)
    for k, v in losses.items():
        arrays["loss:" + k] = np.asarray(float(torch.mean(v)), dtype=np.float64)
    arrays["loss:total"] = np.asarray(float(total), dtype=np.float64)
    for k in keys:
        if k in grads:
            arrays["grad:" + k] = grads[k].numpy().astype(np.float32)
    arrays["gradnorm"] = np.asarray([float(g.norm()) for g in grads.values()], dtype=np.float64)
    arrays["gradnames"] = np.asarray(list(grads.keys()))
    arrays["shapes"] = np.asarray(json.dumps({k: list(s) for k, s in shapes.items()}))
    arrays["meta"] = np.asarray(json.dumps(dict(case=name, cfg=c, torch=torch.__version__,
                                                  transformers=transformers.__version__,
                                                  reference="zeyun-zhong/AFFT @ /root/reference (v1)")))
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **arrays)


def run_full_case(name: str, c: dict):
    """A full-size fixture (cases.FULL_CASES): the reference's forward + loss + backward at the real widths on closed-form
    weights, cross-checked against the oracle, stored compactly."""
    import closed_form as cf
    from cases import FULL_GRAD_SAMPLES, oracle_cfg
    from helpers import compact_entry, strided_sample
    from models.base_model import BaseModel
    from common.runner import BasicLossAccuracy, Runner
    from oracle import afft_oracle as O

    torch.manual_seed(0)
    with CudaToCpu():
        model = BaseModel(build_model_cfg(c), num_classes={"action": c["num_classes"]}, class_mappings={})
    model.eval()
    sd = model.state_dict()
    shapes = {k: tuple(v.shape) for k, v in sd.items() if v.dtype.is_floating_point and not k.endswith(".attn.bias")
              and not k.endswith("masked_bias")}
    state = cf.fill_state(shapes)
    missing = model.load_state_dict(state, strict=False)
    assert not missing.unexpected_keys, missing
    B, T, K = c["B"], c["T"], c["num_classes"]
    data = cf.inputs_for(name, c["modal_dims"], B, T)
    tgt, sub = cf.labels_for(name, B, T, K, c.get("ignore_frac", 0.25))
    with CudaToCpu():
        outputs, out_t = model({m: d.clone() for m, d in data.items()}, mixup_fn=None, target={"action": tgt},
                               target_subclips={"action": sub}, target_subclips_ignore_index=None)
        losses, _ = BasicLossAccuracy()(outputs, out_t["target"], out_t["target_subclips"], mixup_enable=False,
                                         target_subclips_ignore_index=out_t["target_subclips_ignore_index"])
    total, _ = Runner._reduce_loss(losses, {"cls_action": 1.0, "past_cls_action": 1.0, "past_reg": 1.0})
    model.zero_grad()
    total.backward()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}

    P = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    oout = O.base_model_forward(P, {m: d.clone() for m, d in data.items()}, oracle_cfg(c))
    ototal, _ = O.loss(oout, tgt, sub)
    ototal.backward()
    ref_flat, ora_flat = flatten_outputs(outputs), flatten_outputs(oout)
    worst = 0.0
    for k, v in ref_flat.items():
        if k == "attentions/modality_attns" and c["fuser"] == "ca":
            continue
        e = float((ora_flat[k].detach() - v.detach()).norm() / (v.detach().norm() + 1e-30))
        worst = max(worst, e)
        assert e < 2e-5, (name, k, e)
    assert abs(float(ototal) - float(total)) < 2e-5 * max(1.0, abs(float(total)))
    gmed = float(np.median([float(g.norm()) for g in grads.values()]))
    for k, g in grads.items():
        e = float((P[k].grad - g).norm() / max(float(g.norm()), 1e-6 * gmed))
        worst = max(worst, e)
        assert e < 1e-4, (name, "grad", k, e)
    print(f"[{name}] oracle == reference at full size: worst rel-L2 {worst:.2e}; loss {float(total):.6f}; {len(grads)} gradients")

    arrays = {}
    for k, v in ref_flat.items():
        if k == "attentions/modality_attns" and c["fuser"] == "ca":
            continue
        for kk, a in compact_entry(v).items():
            arrays[f"out:{k}:{kk}"] = a
    for k, v in losses.items():
        arrays["loss:" + k] = np.asarray(float(torch.mean(v)), dtype=np.float64)
    arrays["loss:total"] = np.asarray(float(total), dtype=np.float64)
    names = list(grads.keys())
    samples = [strided_sample(grads[k].double(), FULL_GRAD_SAMPLES).float().numpy() for k in names]
    arrays["gradnames"] = np.asarray(names)
    arrays["gradnorm"] = np.asarray([float(grads[k].double().norm()) for k in names], dtype=np.float64)
    arrays["gradsamples"] = np.concatenate(samples)
    arrays["gradsample_offsets"] = np.cumsum([0] + [len(x) for x in samples]).astype(np.int64)
    arrays["shapes"] = np.asarray(json.dumps({k: list(s) for k, s in shapes.items()}))
    arrays["meta"] = np.asarray(json.dumps(dict(case=name, cfg=c, torch=torch.__version__, transformers=transformers.__version__,
                                                  parameters=int(sum(int(np.prod(s)) for s in shapes.values())),
                                                  reference="zeyun-zhong/AFFT @ /root/reference (v1)")))
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **arrays)


def run_eval_case():
    """m0_marginalize: the reference's challenge.marginalize_verb_noun (challenge.py:196-210) + its accuracy bookkeeping
    (compute_accuracies_epic :161-193, common/utils.py:19-56) on a stub dataset object."""
    import pandas as pd
    _mod("h5py")
    _mod("numpyencoder", NumpyEncoder=object)
    import challenge as RC
    from oracle import afft_oracle as O
    from closed_form import eval_inputs
    logits, mv, mn, a_lab, v_lab, n_lab = eval_inputs()

    class _DS:
        class_mappings = {("verb", "action"): torch.from_numpy(mv), ("noun", "action"): torch.from_numpy(mn)}
        df = pd.DataFrame(dict(verb_class=v_lab, noun_class=n_lab, action_class=a_lab))
        classes_manyshot = {}
        version = RC.EPIC100_VERSION
    acc, scores = RC.marginalize_verb_noun(logits.copy(), _DS, to_prob=True)
    oacc, oscores = O.marginalize_verb_noun(logits.copy(), mv, mn, v_lab, n_lab, a_lab)
    for r, o in zip(scores, oscores):
        assert np.allclose(r, o, rtol=1e-6, atol=1e-7)
    for k, v in acc.items():
        assert (np.isnan(v) and np.isnan(oacc[k])) or abs(v - oacc[k]) < 1e-9, (k, v, oacc[k])
    print("[m0_marginalize] oracle == reference:", {k: round(float(v), 3) for k, v in acc.items() if not np.isnan(v)})
    np.savez_compressed(os.path.join(HERE, "m0_marginalize.npz"), verb=np.asarray(scores[0], np.float32),
                        noun=np.asarray(scores[1], np.float32), action=np.asarray(scores[2], np.float32),
                        acc_names=np.asarray(sorted(acc)), acc_values=np.asarray([float(acc[k]) for k in sorted(acc)], np.float64),
                        meta=np.asarray(json.dumps(dict(case="m0_marginalize", torch=torch.__version__, numpy=np.__version__,
                                                        reference="zeyun-zhong/AFFT @ /root/reference (v1)"))))


def run_unseen_tail_case():
    """m1_unseen_tail: the reference's compute_accuracies_epic(..., compute_manyshot_unseen_tail=True) on an EPIC-100 stub
    (challenge.py:109-193): many-shot, unseen-participant and tail-class recalls.  The reference calls
    pd.read_csv(..., squeeze=True), a keyword pandas 2 removed; for this run pd.read_csv is wrapped to give the keyword its
    pandas-1 meaning (a one-column frame comes back as its Series) -- nothing of the reference itself is changed."""
    import tempfile
    import pandas as pd
    _mod("h5py")
    _mod("numpyencoder", NumpyEncoder=object)
    import challenge as RC
    from closed_form import eval_inputs, unseen_tail_tables
    logits, mv, mn, a_lab, v_lab, n_lab = eval_inputs()
    ids, tables, manyshot = unseen_tail_tables(len(a_lab), logits.shape[1])
    z = np.load(os.path.join(HERE, "m0_marginalize.npz"))
    scores = [z["verb"], z["noun"], z["action"]]
    plain_read_csv = pd.read_csv

    def read_csv(*a, squeeze=False, **kw):
        out = plain_read_csv(*a, **kw)
        return out.squeeze("columns") if squeeze else out
    with tempfile.TemporaryDirectory() as d:
        for fn, rows in tables.items():
            with open(os.path.join(d, fn), "w") as fh:
                fh.write("\n".join(rows) + "\n")

        class _DS:
            df = pd.DataFrame(dict(verb_class=v_lab, noun_class=n_lab, action_class=a_lab, narration_id=ids))
            classes_manyshot = manyshot
            version = RC.EPIC100_VERSION
            rulstm_annotation_dir = d
        RC.pd.read_csv = read_csv
        try:
            acc = RC.compute_accuracies_epic(scores, _DS, compute_manyshot_unseen_tail=True)
        finally:
            RC.pd.read_csv = plain_read_csv
    print("[m1_unseen_tail]", {k: round(float(v), 3) for k, v in acc.items()})
    assert len(acc) == 18 and not any(np.isnan(float(v)) for v in acc.values())
    np.savez_compressed(os.path.join(HERE, "m1_unseen_tail.npz"), acc_names=np.asarray(sorted(acc)),
                        acc_values=np.asarray([float(acc[k]) for k in sorted(acc)], np.float64),
                        meta=np.asarray(json.dumps(dict(case="m1_unseen_tail", scores="m0_marginalize.npz", pandas=pd.__version__,
                                                        reference="zeyun-zhong/AFFT @ /root/reference (v1)"))))


def run_reader_case():
    """r0_reader: the reference's EpicRULSTMFeatsReader (datasets/reader_fns.py:41-157) over dict-backed fake LMDB
    environments (the `lmdb` / torchvision packages are stubbed; the reader only calls env.begin().get(key))."""
    from closed_form import reader_stores
    vid, stores, queries = reader_stores()

    class _Txn:
        def __init__(self, d):
            self.d = d

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def get(self, k):
            return self.d.get(k)

    class _Env:
        def __init__(self, d):
            self.d = d

        def begin(self):
            return _Txn(self.d)

    _mod("lmdb", open=lambda path, readonly=True, lock=False: _Env(stores["audio" if "audio" in str(path) else "rgb"]))
    _mod("torchvision")
    sys.modules["omegaconf"].OmegaConf.get_type = staticmethod(lambda x: type(x))
    import datasets.reader_fns as RR
    import logging
    logging.disable(logging.CRITICAL)
    out = {}
    for tag, paths in (("rgb", ["/data/rgb_lmdb"]), ("rgb_audio", ["/data/rgb_lmdb", "/data/audio_lmdb"])):
        rd = RR.EpicRULSTMFeatsReader(lmdb_path=paths, warn_if_using_closeby_frame=True)
        for qi, (a, b) in enumerate(queries):
            feat, _, _, _ = rd(f"/videos/{vid}.MP4", a, b, 30.0, None)
            out[f"{tag}:{qi}"] = feat.numpy()
    logging.disable(logging.NOTSET)
    np.savez_compressed(os.path.join(HERE, "r0_reader.npz"), **out,
                        meta=np.asarray(json.dumps(dict(case="r0_reader", reference="zeyun-zhong/AFFT @ /root/reference (v1)"))))
    print("[r0_reader]", {k: v.shape for k, v in out.items()})


def run_edge_case():
    """e0_edges: the reference's transformerblock.Block with an ARBITRARY additive attention mask (models/transformerblock.py:26-28 adds
    whatever tensor it is given) and DecoderBlock(dim, mem_dim != dim, qkv_bias=True) (:41-50, :66-68) -- interface edges the AFFT
    configurations never use.  Closed-form weights and inputs; outputs, attention maps and every gradient of loss = mean(y^2) stored."""
    import closed_form as cf
    from models.transformerblock import Block, DecoderBlock
    from oracle import afft_oracle as O
    N, L, d, dm, H = 3, 6, 64, 128, 4
    out = {}

    def mask(tag):
        m = cf.tensor_for(f"e0.{tag}.mask", (L, L), "input")           # values in [-2, 2)
        kill = cf.tensor_for(f"e0.{tag}.kill", (L, L), "input") > 1.2  # ~ 20 % of the entries: -inf (no row is masked whole: checked)
        kill.fill_diagonal_(False)
        m = m.masked_fill(kill, float("-inf"))
        assert bool(torch.isfinite(m).any(dim=1).all())
        return m

    def fill(mod, tag):
        sd = mod.state_dict()
        state = {k: cf.tensor_for(f"e0.{tag}.{k}", tuple(v.shape)) for k, v in sd.items()}
        mod.load_state_dict(state)
        return state

    # --- Block + arbitrary mask
    blk = Block(d, H).eval()
    st = fill(blk, "block")
    x = cf.tensor_for("e0.block.x", (N, L, d), "input").requires_grad_(True)
    m1 = mask("block")
    y, attn = blk(x, m1)
    (y.pow(2).mean()).backward()
    P = {("b." + k): v.detach().clone().requires_grad_(True) for k, v in st.items()}
    xo = x.detach().clone().requires_grad_(True)
    yo, ao = O.block(P, "b.", xo, H, m1)
    yo.pow(2).mean().backward()
    err = max(float((yo - y).abs().max()), float((ao - attn).abs().max()), float((xo.grad - x.grad).abs().max()))
    assert err < 2e-5, err
    out.update({"block.y": y, "block.attn": attn, "block.dx": x.grad, "block.mask": m1})
    out.update({f"block.grad.{k}": p.grad for k, p in blk.named_parameters()})
    # --- DecoderBlock, mem_dim != dim, qkv_bias, arbitrary mask
    dec = DecoderBlock(d, mem_dim=dm, num_heads=H, qkv_bias=True).eval()
    st2 = fill(dec, "dec")
    x2 = cf.tensor_for("e0.dec.x", (N, L, d), "input").requires_grad_(True)
    mem = cf.tensor_for("e0.dec.mem", (N, L, dm), "input").requires_grad_(True)
    m2 = mask("dec")
    y2 = dec(x2, mem, m2)
    y2.pow(2).mean().backward()
    P2 = {("d." + k): v.detach().clone().requires_grad_(True) for k, v in st2.items()}
    x2o, memo = x2.detach().clone().requires_grad_(True), mem.detach().clone().requires_grad_(True)
    y2o = O.decoder_block(P2, "d.", x2o, memo, H, m2)
    y2o.pow(2).mean().backward()
    err2 = max(float((y2o - y2).abs().max()), float((x2o.grad - x2.grad).abs().max()), float((memo.grad - mem.grad).abs().max()))
    assert err2 < 2e-5, err2
    out.update({"dec.y": y2, "dec.dx": x2.grad, "dec.dmem": mem.grad, "dec.mask": m2})
    out.update({f"dec.grad.{k}": p.grad for k, p in dec.named_parameters()})
    shapes = {"block": {k: list(v.shape) for k, v in st.items()}, "dec": {k: list(v.shape) for k, v in st2.items()}}
    np.savez_compressed(os.path.join(HERE, "e0_edges.npz"), **{k: v.detach().float().numpy() for k, v in out.items()},
                        shapes=np.asarray(json.dumps(shapes)),
                        meta=np.asarray(json.dumps(dict(case="e0_edges", N=N, L=L, d=d, mem_dim=dm, heads=H, torch=torch.__version__,
                                                        reference="zeyun-zhong/AFFT @ /root/reference (v1)"))))
    print("[e0_edges] oracle == reference:", err, err2, "tensors", len(out))


def main():
    install_stubs()
    sys.path.insert(0, os.path.dirname(HERE))       # tests/helpers.py
    from cases import CASES, FULL_CASES
    only = sys.argv[1:]
    for name, c in CASES.items():
        if only and name not in only:
            continue
        run_case(name, c)
    for name, c in FULL_CASES.items():
        if only and name not in only:
            continue
        run_full_case(name, c)
    if not only or "m0_marginalize" in only:
        run_eval_case()
    if not only or "m1_unseen_tail" in only:
        run_unseen_tail_case()
    if not only or "r0_reader" in only:
        run_reader_case()
    if not only or "e0_edges" in only:
        run_edge_case()


if __name__ == "__main__":
    main()

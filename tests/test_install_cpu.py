"""Build-container tests of the drop-in claim: after afft_amd.install_as_models() the reference's import paths and Hydra
`_target_` strings resolve to this package, and every model configuration the reference ships (conf/model/**.yaml, read from
/root/reference when it is present -- the GPU box does not have it) instantiates with the yaml's own keyword arguments."""
import importlib
import itertools
import os
import re
import sys

import pytest
import torch

CONF = "/root/reference/conf/model"


@pytest.fixture()
def installed():
    saved = {k: v for k, v in sys.modules.items() if k == "models" or k.startswith("models.") or k == "common" or k.startswith("common.")}
    for k in saved:
        del sys.modules[k]
    import afft_amd
    afft_amd.install_as_models()
    yield afft_amd
    for k in [k for k in sys.modules if k == "models" or k.startswith("models.") or k == "common" or k.startswith("common.")]:
        del sys.modules[k]
    sys.modules.update(saved)


def test_reference_import_paths_resolve_here(installed):
    import afft_amd.models.fusion as ours
    assert importlib.import_module("models.fusion") is ours
    for name in ("base_model", "fusion", "transformerblock", "future_prediction", "feature_mapping"):
        assert importlib.import_module(f"models.{name}").__name__ == f"afft_amd.models.{name}"
    # registered whether or not `common` had been imported before the call
    for name in ("runner", "mixup", "transforms"):
        assert importlib.import_module(f"common.{name}").__name__ == f"afft_amd.common.{name}"
    from models.fusion import ModalTokenCMFuser, TemporalCrossAttentFuser   # noqa: F401
    from models.future_prediction import BaseFuturePredictor, CMFPEarly     # noqa: F401
    from common.runner import Runner                                        # noqa: F401


def _load(rel):
    import yaml
    with open(os.path.join(CONF, rel)) as f:
        return yaml.safe_load(f)


def _resolve(node, root):
    """${a.b.c} interpolation against `root` (what OmegaConf does for the reference)"""
    if isinstance(node, dict):
        return {k: _resolve(v, root) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve(v, root) for v in node]
    if isinstance(node, str):
        m = re.fullmatch(r"\$\{([\w.]+)\}", node)
        if m:
            cur = root
            for part in m.group(1).split("."):
                cur = cur[part]
            return _resolve(cur, root)
    return node


COMBOS = [(f, m, "cmfp_early") for f, m in itertools.product(["SA-Fuser", "CA-Fuser", "SA-Fuser_wo_token", "T-SA-Fuser"],
                                                               ["linear", "gatedlinear", "nonlinear"])] + \
         [("MATT", "linear", "scorefusion"), ("SA-Fuser", "linear", "individual")]


@pytest.mark.skipif(not os.path.isdir(CONF), reason="reference conf/ not present (GPU box)")
@pytest.mark.parametrize("fuser,mapping,cmfp", COMBOS)
def test_reference_yaml_configs_instantiate(installed, fuser, mapping, cmfp):
    from afft_amd._hydra_compat import to_attr
    modal_dims = {"rgb": 64, "objects": 24, "audio": 64, "flow": 64}
    late = cmfp != "cmfp_early"
    common = _load("common.yaml")
    common.update(backbones={m: _load("backbone/identity.yaml") for m in modal_dims}, fp_inter_dim=128, fp_layers=2, fp_heads=2,
                  modality_cls=late, fusion_cls=not late, share_classifiers=not late)
    model = dict(modal_dims=modal_dims, modal_feature_order=["rgb", "objects", "audio", "poses", "flow"], common_dim=64,
                 dropout=0.2, common=common, mapping=_load(f"mapping/{mapping}.yaml"), fuser=_load(f"fuser/{fuser}.yaml"),
                 future_predictor=_load("future_predictor/base_future_predictor.yaml"), CMFP=_load(f"CMFP/{cmfp}.yaml"))
    if fuser == "T-SA-Fuser":
        model["fuser"]["temporal_sequence_length"] = 4
    cfg = to_attr(_resolve(model, {"model": model}))
    for section in ("fuser", "mapping", "future_predictor", "CMFP"):
        assert cfg[section]["_target_"].startswith("models."), section
    from models.base_model import BaseModel        # the reference's import path (train.py:22)
    net = BaseModel(cfg, num_classes={"action": 11}, class_mappings={})
    assert type(net).__module__ == "afft_amd.models.base_model"
    names = [k for k, _ in net.named_parameters()]
    assert any(k.startswith("future_predictor.") for k in names)
    if not late:
        assert any(".fuser." in k for k in names)
        target = cfg.fuser["_target_"].rsplit(".", 1)[1]
        assert type(net.future_predictor.fuser).__name__ == target

"""afft_amd: MI355X-native (gfx950) implementation of AFFT's data-parallel hot path -- the SA-/CA-Fuser
modality-fusion transformer, the GPT-2 style causal future predictor, the classifier heads and the
three-term anticipation loss, forward and backward -- behind the reference's own module interface.

    afft_amd.models.*          mirrors of the reference's models/{base_model,fusion,transformerblock,
                               future_prediction,feature_mapping}.py
    afft_amd.common.runner     mirror of common/runner.py (loss + step wrapper)
    afft_amd.parallel          data-parallel gradient reduction over RCCL/xGMI + fused SGD
    afft_amd.ops / _lib        the C-ABI (include/afft_hip.h) via ctypes
"""
import sys as _sys

from .runtime import precision, precision_scope, set_grad_mode, set_precision  # noqa: F401


def install_as_models(patch_ddp: bool = False):
    """Make ``import models.fusion`` / Hydra ``_target_: models.fusion.ModalTokenCMFuser`` resolve to this
    package, so the reference's train.py / conf/ work unchanged.
    patch_ddp: also make ``torch.nn.parallel.DistributedDataParallel`` (train.py:364-368) construct
    ``afft_amd.parallel.DistributedDataParallel`` -- the wrapper that works with the gradient sink and hands the all-reduce
    to ``afft_amd.optim.SGD`` (Hydra: ``opt.optimizer._target_=afft_amd.optim.SGD``)."""
    import importlib
    if patch_ddp:
        import torch
        from .parallel import DistributedDataParallel as _DDP
        torch.nn.parallel.DistributedDataParallel = _DDP
    pkg = importlib.import_module("afft_amd.models")
    _sys.modules["models"] = pkg
    for name in ("base_model", "fusion", "transformerblock", "future_prediction", "feature_mapping"):
        _sys.modules[f"models.{name}"] = importlib.import_module(f"afft_amd.models.{name}")
    # common.{runner,mixup,transforms}: registered unconditionally.  When the reference's own `common` package is (or will be)
    # importable its other modules (utils, scheduler, sampler, ...) must keep resolving, so only the three mirrored
    # sub-modules are overridden; without it, this package's `afft_amd.common` stands in as `common`.
    try:
        common = importlib.import_module("common")
    except ImportError:
        common = importlib.import_module("afft_amd.common")
        _sys.modules["common"] = common
    for name in ("runner", "mixup", "transforms"):
        mod = importlib.import_module(f"afft_amd.common.{name}")
        _sys.modules[f"common.{name}"] = mod
        setattr(common, name, mod)

"""hydra / omegaconf shims.  The reference builds its modules with ``hydra.utils.instantiate`` from
``_target_`` strings (models/base_model.py:22-25, models/future_prediction.py:51,76,85-93) and checks
``isinstance(model_cfg.modal_dims, DictConfig)``.  When hydra/omegaconf are installed they are used as is;
this image has neither, so a minimal non-recursive ``instantiate`` and an attribute-dict stand in.
``_target_`` paths under ``models.`` resolve to this package's mirrors (``afft_amd.models.*``)."""
from __future__ import annotations

import importlib

try:  # pragma: no cover - not available in the build image
    from omegaconf import DictConfig as _OmegaDictConfig  # type: ignore
except Exception:  # noqa: BLE001
    _OmegaDictConfig = None

try:  # pragma: no cover
    from hydra.utils import instantiate as _hydra_instantiate  # type: ignore
except Exception:  # noqa: BLE001
    _hydra_instantiate = None


class AttrDict(dict):
    """dict with attribute access (the slice of DictConfig behaviour the path relies on)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def to_attr(o):
    if isinstance(o, dict) and not isinstance(o, AttrDict):
        return AttrDict({k: to_attr(v) for k, v in o.items()})
    return o


def is_dict_config(o) -> bool:
    if _OmegaDictConfig is not None and isinstance(o, _OmegaDictConfig):
        return True
    return isinstance(o, dict)


def _resolve(target: str):
    modname, cls = target.rsplit(".", 1)
    candidates = [modname]
    if modname == "models" or modname.startswith("models."):
        candidates.insert(0, "afft_amd." + modname)
    last = None
    for m in candidates:
        try:
            return getattr(importlib.import_module(m), cls)
        except (ImportError, AttributeError) as e:  # noqa: PERF203
            last = e
    raise ImportError(f"cannot resolve _target_ {target!r}: {last}")


def instantiate(cfg, *args, **kwargs):
    """Non-recursive instantiate: ``cfg['_target_'](*args, **{**cfg, **kwargs})``."""
    if _hydra_instantiate is not None and _OmegaDictConfig is not None and isinstance(cfg, _OmegaDictConfig):
        return _hydra_instantiate(cfg, *args, **kwargs)
    kwargs.pop("_recursive_", None)
    params = {k: v for k, v in dict(cfg).items() if k not in ("_target_", "_recursive_")}
    params.update(kwargs)
    return _resolve(cfg["_target_"])(*args, **params)

"""Restatement of the two learning-rate schedules the reference's training recipes use (``common/scheduler.py:57-160``:
``opt/scheduler=cosine`` wrapped in ``opt.warmup``; expts/01: 20 warm-up + 30 cosine epochs), stepped once per ITERATION
(train.py:264-265), for environments where the reference's own ``common`` package is not importable (the GPU box, the
tests) and for ``bench.py``'s reference-loop measurement.  With the reference on the path its own module is used
(``afft_amd.install_as_models`` never replaces ``common.scheduler``).

Both are ``torch.optim.lr_scheduler.LRScheduler`` subclasses, so they drive any ``torch.optim.Optimizer`` -- in particular
``afft_amd.optim.SGD``, which reads ``param_groups[i]['lr']`` at the start of every backward pass.
"""
from __future__ import annotations

import math

import torch
from torch.optim.lr_scheduler import LRScheduler


class CosineLR(LRScheduler):
    """Cosine annealing from each group's base lr down to ``eta_min * world_size`` over ``num_epochs * iters_per_epoch``
    iterations, 0 afterwards (common/scheduler.py:57-76).  The closed form of torch's CosineAnnealingLR, which the reference
    subclasses; its recursive form and this one agree to rounding."""

    def __init__(self, optimizer, num_epochs, iters_per_epoch=None, world_size=None, eta_min=0.0, last_epoch=-1):
        self.T_max = int(num_epochs * iters_per_epoch)
        self.eta_min = float(eta_min) * (world_size or 1)
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        t = self.last_epoch
        if t >= self.T_max:
            return [0.0 for _ in self.base_lrs]
        c = 0.5 * (1.0 + math.cos(math.pi * t / self.T_max))
        return [self.eta_min + (b - self.eta_min) * c for b in self.base_lrs]


class Warmup(LRScheduler):
    """Linear warm-up from ``init_lr_ratio * base_lr`` to ``base_lr`` over ``num_epochs * iters_per_epoch`` iterations, then
    every ``step()`` is the wrapped scheduler's (common/scheduler.py:87-139)."""

    def __init__(self, optimizer, scheduler, init_lr_ratio: float = 0.0, num_epochs: int = 5, last_epoch: int = -1,
                 iters_per_epoch: int = None, world_size: int = None):
        self.base_scheduler = scheduler
        self.warmup_iters = max(int(num_epochs * iters_per_epoch), 1)
        self.init_lr_ratio = init_lr_ratio if self.warmup_iters > 1 else 1.0     # no 0 -> 1 jump inside one iteration
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        assert self.last_epoch < self.warmup_iters
        f = self.init_lr_ratio + (1.0 - self.init_lr_ratio) * (float(self.last_epoch) / self.warmup_iters)
        return [b * f for b in self.base_lrs]

    def step(self, *args, **kwargs):
        if self.last_epoch < self.warmup_iters - 1:
            super().step(*args, **kwargs)
        else:
            self.base_scheduler.step(*args, **kwargs)

    def state_dict(self):
        other = {k: v for k, v in self.__dict__.items() if k not in ("base_scheduler", "optimizer")}
        return {"base_sched_dict": self.base_scheduler.state_dict(), "other_stuff": other}

    def load_state_dict(self, state_dict):
        self.base_scheduler.__dict__.update(state_dict["base_sched_dict"])
        self.__dict__.update(state_dict["other_stuff"])


def prepare_params(model: torch.nn.Module, lr_wd, overall_lr: float, overall_wd: float):
    """The per-parameter optimizer groups of train.py:189-225: one group per named parameter (``lr_wd``: [[module names, lr,
    wd], ...] overrides for sub-modules, ``'__all__'`` = the model); groups with lr = 0 are dropped and their parameters
    frozen."""
    import operator
    named = dict(model.named_parameters())
    groups, rest = [], dict(named)
    for module_names, lr, wd in (lr_wd or []):
        if not isinstance(module_names, (list, tuple)):
            module_names = [module_names]
        chosen = {}
        for mn in module_names:
            mod = model if mn == "__all__" else operator.attrgetter(mn)(model)
            chosen.update({mn + "." + n: p for n, p in mod.named_parameters()})
        groups += [{"params": p, "lr": lr, "weight_decay": wd, "name": n} for n, p in chosen.items()]
        rest = {n: p for n, p in rest.items() if n not in chosen}
    groups += [{"params": p, "lr": overall_lr, "weight_decay": overall_wd, "name": n} for n, p in rest.items()]
    out = []
    for g in groups:
        if g["lr"] != 0.0:
            out.append(g)
        else:
            g["params"].requires_grad = False
    return out

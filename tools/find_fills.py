"""Which Python lines launch the small torch kernels (fill / zero / cat / add) inside a training step?  torch.profiler with stacks."""
import os
import sys
from collections import Counter

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
import afft_amd  # noqa: E402
from afft_amd.parallel import Trainer  # noqa: E402

afft_amd.set_precision("bf16")
afft_amd.set_grad_mode("sink")
dev = torch.device("cuda:0")
model, c = B.build_model("cfg2", dev)
feats, tgt, sub = B.make_inputs(c, 64, c["T"], 0, dev)
tr = Trainer(model, {"cls_action": 1.0, "past_cls_action": 1.0, "past_reg": 1.0})
model.train()
for _ in range(5):
    tr.step(feats, tgt, sub)
torch.cuda.synchronize()
import traceback
cnt = Counter()
_zeros, _zero_, _fill_, _zl = torch.zeros, torch.Tensor.zero_, torch.Tensor.fill_, torch.zeros_like


def _where():
    fr = [f for f in traceback.extract_stack()[:-2] if "afft_amd" in f.filename or f.filename.endswith("bench.py")][-2:]
    return " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(fr))


def zeros(*a, **k):
    cnt[("torch.zeros", _where())] += 1
    return _zeros(*a, **k)


def zeros_like(*a, **k):
    cnt[("torch.zeros_like", _where())] += 1
    return _zl(*a, **k)


def zero_(self):
    cnt[("Tensor.zero_", _where())] += 1
    return _zero_(self)


def fill_(self, v):
    cnt[("Tensor.fill_", _where())] += 1
    return _fill_(self, v)


with torch.autograd.set_multithreading_enabled(False):
    tr.step(feats, tgt, sub)
    torch.zeros, torch.zeros_like, torch.Tensor.zero_, torch.Tensor.fill_ = zeros, zeros_like, zero_, fill_
    tr.step(feats, tgt, sub)
    torch.zeros, torch.zeros_like, torch.Tensor.zero_, torch.Tensor.fill_ = _zeros, _zl, _zero_, _fill_
torch.cuda.synchronize()
for (name, where), n in cnt.most_common(40):
    print(f"{n:4d} {name:18s} {where}")

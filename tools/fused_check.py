"""Full-size check of the fused optimizer epilogue: relative difference of parameters / momentum after N steps between
(a) two runs of the separate update (0: every reduction of the path is ordered) and (b) fused vs separate (0 as well).
usage: python tools/fused_check.py [config] [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import afft_amd  # noqa: E402
from afft_amd import dropout as D_, runtime as rt  # noqa: E402
from afft_amd.config import BASELINE_CONFIGS, make_model_cfg  # noqa: E402
from afft_amd.models.base_model import BaseModel  # noqa: E402
from afft_amd.parallel import Trainer  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
c = BASELINE_CONFIGS[name]
B, T, K = 64, c["T"], 3806
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(8)
feats = {m: torch.randn(B, T, C, 1, 1, 1, generator=g).to(dev) for m, C in c["modal_dims"].items()}
tgt = {"action": torch.randint(0, K, (B,), generator=g).to(dev)}
sub = {"action": torch.randint(0, K, (B, T, 1), generator=g).to(dev)}
wts = {"cls_action": 1.0, "past_cls_action": 1.0, "past_reg": 1.0}


def run(fused, drop):
    rt.set_fused_sgd(fused)
    D_.manual_seed(17)
    torch.manual_seed(9)
    cfg = make_model_cfg(c["modal_dims"], c["common_dim"], c["fp_inter_dim"], fuser=c["fuser"], T=T, drop=drop)
    model = BaseModel(cfg, {"action": K}, {}).to(dev).train()
    tr = Trainer(model, wts, lr=1e-2)
    p0 = tr.flat.flat_p.clone()
    for _ in range(steps):
        tr.step(feats, tgt, sub)
    torch.cuda.synchronize()
    out = (tr.flat.flat_p.clone(), tr.opt.buf.clone(), p0)
    del tr, model
    torch.cuda.empty_cache()
    return out


def rel(a, b, base=None):
    d = (a.double() - b.double()).norm()
    n = (b.double() - base.double()).norm() if base is not None else b.double().norm()
    return float(d / n)


for drop in (0.0, 0.1):
    s1, s2, f1 = run(False, drop), run(False, drop), run(True, drop)
    print(f"{name} drop={drop} steps={steps}: separate vs separate  dp/|update| {rel(s1[0], s2[0], s1[2]):.3e}  momentum {rel(s1[1], s2[1]):.3e}")
    print(f"{name} drop={drop} steps={steps}: fused    vs separate  dp/|update| {rel(f1[0], s1[0], s1[2]):.3e}  momentum {rel(f1[1], s1[1]):.3e}")

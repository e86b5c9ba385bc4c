"""Which composite backward calls do NOT get their bf16 dy handed over by the LayerNorm backward downstream (and so run a cast kernel)?
usage: python tools/shadow_diag.py [config]"""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
import afft_amd
from afft_amd import functional as F_
from afft_amd.parallel import Trainer
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
afft_amd.set_precision("bf16")
dev = torch.device("cuda:0")
model, c = B.build_model(name, dev)
feats, tgt, sub = B.make_inputs(c, 64, c["T"], 0, dev)
tr = Trainer(model, {"cls_action": 1.0, "past_cls_action": 1.0, "past_reg": 1.0})
model.train()
for _ in range(3):
    tr.step(feats, tgt, sub)
log = collections.Counter()
orig = F_._take_shadow
def spy(dy, od, bias):
    sh = F_._TS.shadow
    why = "ok"
    if sh is None:
        why = "no shadow published"
    elif sh.ptr != dy.data_ptr():
        why = "other tensor (autograd summed / copied the gradient)"
    elif sh.shape != tuple(dy.shape):
        why = "shape"
    elif sh.version != dy._version:
        why = "version"
    elif sh.bias is not bias:
        why = "bias"
    r = orig(dy, od, bias)
    log[(tuple(dy.shape), why if r is None else "accepted")] += 1
    return r
F_._take_shadow = spy
tr.step(feats, tgt, sub)
torch.cuda.synchronize()
for k, v in sorted(log.items(), key=lambda kv: -kv[1]):
    print(v, k)

"""Soak: N training steps of the bench workload on one fixed batch (the loss must fall, stay finite, memory must not grow).
usage: python tools/soak.py [precision] [steps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
import afft_amd
from afft_amd.parallel import Trainer
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
afft_amd.set_precision(prec)
dev = torch.device("cuda:0")
model, c = B.build_model("cfg2", dev)
feats, tgt, sub = B.make_inputs(c, 64, c["T"], 0, dev)
tr = Trainer(model, {"cls_action": 1.0, "past_cls_action": 1.0, "past_reg": 1.0}, lr=1e-3)
model.train()
mem0 = None
for i in range(steps):
    loss, parts = tr.step(feats, tgt, sub)
    if i % 50 == 0 or i == steps - 1:
        torch.cuda.synchronize()
        mem = torch.cuda.memory_allocated() / 2 ** 20
        if i == 50:
            mem0 = mem
        print(f"{prec} step {i:4d} loss {float(loss):.4f}  allocated {mem:.0f} MiB  ok {float(tr.opt.ok)}", flush=True)
assert float(loss) == float(loss)
if mem0 is not None:
    assert mem < mem0 * 1.02 + 64, (mem0, mem)
print("soak ok")

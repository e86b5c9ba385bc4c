"""Host enqueue time vs GPU time of one training step (is the step launch-bound?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
import afft_amd
from afft_amd.parallel import Trainer
afft_amd.set_precision("bf16")
dev = torch.device("cuda:0")
model, c = B.build_model("cfg2", dev)
feats, tgt, sub = B.make_inputs(c, 64, c["T"], 0, dev)
tr = Trainer(model, {"cls_action": 1.0, "past_cls_action": 1.0, "past_reg": 1.0})
model.train()
for _ in range(3):
    tr.step(feats, tgt, sub)
torch.cuda.synchronize()
host, total = [], []
for _ in range(8):
    t0 = time.perf_counter()
    tr.step(feats, tgt, sub)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
print("host enqueue ms", sorted(host)[len(host)//2], "step ms (enqueue+drain)", sorted(total)[len(total)//2])

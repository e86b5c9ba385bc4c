"""Which torch-native (at::native) kernels a training step still launches, with the Python line that asked for them
(torch.profiler with stacks over 3 steps of the bench workload).  usage: python tools/step_ops.py [config] [batch]"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
import afft_amd  # noqa: E402
from afft_amd.parallel import Trainer  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
afft_amd.set_precision("bf16")
dev = torch.device("cuda:0")
model, c = B.build_model(name, dev)
feats, tgt, sub = B.make_inputs(c, batch, c["T"], 0, dev)
tr = Trainer(model, {"cls_action": 1.0, "past_cls_action": 1.0, "past_reg": 1.0})
model.train()
for _ in range(4):
    tr.step(feats, tgt, sub)
torch.cuda.synchronize()
N = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    for _ in range(N):
        tr.step(feats, tgt, sub)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for ev in prof.events():
    if ev.device_type.name != "CPU" or not ev.name.startswith("aten::"):
        continue
    dev_us = sum(k.duration for k in ev.kernels) if ev.kernels else 0.0
    if not ev.kernels:
        continue
    stack = [s for s in (ev.stack or []) if ("afft_amd/" in s or "bench.py" in s) and "tools/" not in s]
    site = " <- ".join(x.split("afft_amd/")[-1] for x in stack[:2]) if stack else "?"
    agg[(ev.name, site)][0] += len(ev.kernels)
    agg[(ev.name, site)][1] += dev_us
print(f"{name} B={batch}: torch-native ops that launch kernels, per step (over {N} steps)")
tot_n = tot_us = 0
for (op, site), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n / N:6.1f} launches {us / N:8.1f} us  {op:16} {site[:150]}")
    tot_n += n / N
    tot_us += us / N
print(f"total: {tot_n:.1f} launches, {tot_us:.1f} us of device time per step")

"""How far ahead of the GPU does the host run?  Per training step: host time inside Trainer.step (enqueue), and the lead of the
host over the GPU at the step boundary (GPU completion time of step i minus the host time at which step i's enqueue returned).
A lead near zero means something in the step waits for the GPU.  usage: python tools/host_ahead.py [config] [batch] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
import afft_amd  # noqa: E402
from afft_amd.parallel import Trainer  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n = int(sys.argv[3]) if len(sys.argv) > 3 else 30
afft_amd.set_precision("bf16")
afft_amd.set_grad_mode("sink")
dev = torch.device("cuda:0")
model, c = B.build_model(name, dev)
feats, tgt, sub = B.make_inputs(c, batch, c["T"], 0, dev)
tr = Trainer(model, {"cls_action": 1.0, "past_cls_action": 1.0, "past_reg": 1.0})
model.train()
for _ in range(5):
    tr.step(feats, tgt, sub)
torch.cuda.synchronize()
ev0 = torch.cuda.Event(enable_timing=True)
ev0.record()
t_base = time.perf_counter()
evs, host = [], []
for i in range(n):
    t0 = time.perf_counter()
    tr.step(feats, tgt, sub)
    t1 = time.perf_counter()
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    evs.append(e)
    host.append((t0 - t_base, t1 - t_base))
torch.cuda.synchronize()
gpu_done = [ev0.elapsed_time(e) for e in evs]
print(f"{name} B={batch}: step  host_enqueue_ms  host_return_at_ms  gpu_done_at_ms  host_lead_ms")
for i in range(n):
    print(f"  {i:3d}  {(host[i][1] - host[i][0]) * 1e3:8.2f}  {host[i][1] * 1e3:10.2f}  {gpu_done[i]:10.2f}  {gpu_done[i] - host[i][1] * 1e3:8.2f}")
enq = sorted((h[1] - h[0]) * 1e3 for h in host)
print(f"host enqueue p50 {enq[n // 2]:.2f} ms; GPU ms/step {(gpu_done[-1] - gpu_done[4]) / (n - 5):.2f}")

"""Eval-mode forwards of a bench configuration in a loop (for rocprofv3 --kernel-trace + tools/timeline.py with the once-per-
forward marker `assemble`) and the wall-clock p50.  usage: python tools/fwd_loop.py [config] [batch] [n]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import afft_amd  # noqa: E402
from afft_amd.config import BASELINE_CONFIGS, make_model_cfg  # noqa: E402
from afft_amd.models.base_model import BaseModel  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n = int(sys.argv[3]) if len(sys.argv) > 3 else 60
c = BASELINE_CONFIGS[name]
dev = torch.device("cuda:0")
afft_amd.set_precision("bf16")
torch.manual_seed(42)
cfg = make_model_cfg(c["modal_dims"], c["common_dim"], c["fp_inter_dim"], fuser=c["fuser"], T=c["T"])
model = BaseModel(cfg, {"action": 3806}, {}).to(dev).eval()
feats = {m: torch.randn(B, c["T"], C, 1, 1, 1, device=dev) for m, C in c["modal_dims"].items()}
tgt = {"action": torch.randint(0, 3806, (B,), device=dev)}
sub = {"action": torch.randint(0, 3806, (B, c["T"], 1), device=dev)}
ts = []
with torch.no_grad():
    for i in range(n + 5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model(feats, mixup_fn=None, target=tgt, target_subclips=sub, target_subclips_ignore_index=None)
        t1 = time.perf_counter()          # host enqueue done
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if i >= 5:
            ts.append((t2 - t0, t1 - t0))
ts.sort()
print(f"{name} B={B}: forward p50 {ts[len(ts) // 2][0] * 1e3:.3f} ms; host enqueue p50 {sorted(t[1] for t in ts)[len(ts) // 2] * 1e3:.3f} ms")

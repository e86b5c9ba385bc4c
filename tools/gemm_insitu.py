"""Per-shape durations of the bf16 GEMM launches INSIDE a training step (the library's trace hook: a HIP event pair around every
launch on its own stream), to set beside the stand-alone numbers of tools/gemm_bench.py.  usage: python tools/gemm_insitu.py [config] [batch]"""
import os
import sys
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
import afft_amd  # noqa: E402
from afft_amd.parallel import Trainer  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
afft_amd.set_precision(os.environ.get("PRECISION", "bf16"))      # PRECISION=fp16x2: the parity-grade step
afft_amd.set_grad_mode("sink")
dev = torch.device("cuda:0")
model, c = B.build_model(name, dev)
feats, tgt, sub = B.make_inputs(c, batch, c["T"], 0, dev)
tr = Trainer(model, {"cls_action": 1.0, "past_cls_action": 1.0, "past_reg": 1.0})
model.train()
for _ in range(6):
    tr.step(feats, tgt, sub)
torch.cuda.synchronize()
agg = defaultdict(lambda: [0, 0.0])
EVAL = os.environ.get("EVAL") == "1"        # EVAL=1: eval-mode forwards only (no pre-activation stores, nothing beside them)
if EVAL:
    model.eval()
for _ in range(5):
    with B.GemmTimer() as gt:
        if EVAL:
            with torch.no_grad():
                model(feats, mixup_fn=None, target=tgt, target_subclips=sub, target_subclips_ignore_index=None)
        else:
            tr.step(feats, tgt, sub)
    for r in gt.records:
        lay = ("t" if r.a_kstrided else "n") + ("n" if r.b_kstrided else "t")
        k = (lay, r.M, r.N, r.K, r.variant, r.splitk, r.fused_update, int(r.split3))
        agg[k][0] += 1
        agg[k][1] += r.ms
print(f"{name} B={batch}: {'eval-mode forward' if EVAL else 'in-step'} GEMM launches, 5 instrumented {'forwards' if EVAL else 'steps'} (layout: nt forward of nn.Linear / dgrad of Conv1D, nn dgrad of nn.Linear / forward of Conv1D, tn weight gradient)")
print(f"{'layout':6} {'M':>6} {'N':>6} {'K':>6} {'tile':>5} {'splitK':>6} {'fusedSGD':>8} {'split3':>6} {'n/step':>6} {'avg us':>8} {'TFLOP/s':>8}")
tot = 0.0
for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lay, M, N, K, var, sk, fu, x3 = k
    avg = ms / n
    tot += ms / 5
    print(f"{lay:6} {M:6d} {N:6d} {K:6d} {({3: '256', 13: '256s', 10: 'bd160', 9: 'bd256', 8: 'bd160', 7: 'bd256'}.get(var, '128')):>5} {sk:6d} {fu:8d} {x3:6d} {n / 5:6.1f} {avg * 1e3:8.1f} {2.0 * M * N * K / avg / 1e9:8.1f}")
print(f"sum of GEMM launch durations per step: {tot:.2f} ms")

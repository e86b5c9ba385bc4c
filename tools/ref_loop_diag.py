"""Where the reference-shaped loop (bench.py: reference_loop.afft_sgd) spends more than Trainer.step: the same model stepped
with pieces of the loop switched on one at a time.  usage: python tools/ref_loop_diag.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
import afft_amd  # noqa: E402
from afft_amd import dropout as D_  # noqa: E402
from afft_amd.common.mixup import MixUp  # noqa: E402
from afft_amd.common.runner import Runner  # noqa: E402
from afft_amd.common.scheduler import CosineLR, Warmup, prepare_params  # noqa: E402
from afft_amd.optim import SGD  # noqa: E402
from afft_amd.parallel import Trainer  # noqa: E402

dev = torch.device("cuda:0")
wts = {"cls_action": 1.0, "past_cls_action": 1.0, "past_reg": 1.0}


def run(kind):
    D_.manual_seed(42)
    afft_amd.set_grad_mode("sink")
    model, c = B.build_model("cfg2", dev)
    feats, tgt, sub = B.make_inputs(c, 64, c["T"], 0, dev)
    model.train()
    mix = MixUp(alpha=0.1, label_smoothing={"action": 0.4}, num_classes={"action": 3806})
    batch = ({"data_dict": feats, "target": tgt, "target_subclips": sub}, {})
    host = []
    if kind.startswith("trainer"):
        tr = Trainer(model, wts)
        m = mix if "mixup" in kind else None
        step = lambda: tr.step(feats, tgt, sub, mixup_fn=m)      # noqa: E731
    else:
        opt = SGD(prepare_params(model, None, 1e-3, 1e-6), lr=1e-3, momentum=0.9, nesterov=True)
        sched = Warmup(opt, CosineLR(opt, num_epochs=30, iters_per_epoch=1000, world_size=1, eta_min=1e-6), init_lr_ratio=0.01,
                       num_epochs=20, iters_per_epoch=1000, world_size=1) if "sched" in kind else None
        runner = Runner(model, dev, wts, compute_metrics=("metrics" in kind))
        m = mix if "mixup" in kind else None

        def step():
            t0 = time.perf_counter()
            loss, metrics = runner(batch, m, True)
            t1 = time.perf_counter()
            opt.zero_grad()
            loss.backward()
            t2 = time.perf_counter()
            opt.step()
            if sched is not None:
                sched.step()
            t3 = time.perf_counter()
            if "consume" in kind:
                for k, v in metrics.items():
                    if isinstance(v, dict):
                        for a in v.values():
                            float(__import__("numpy").asarray(a).reshape(-1)[0])
                    elif not isinstance(v, torch.Tensor):
                        float(v)
            host.append((t1 - t0, t2 - t1, t3 - t2, time.perf_counter() - t3))
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    host.clear()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10 * 1e3
    h = [round(sum(x[i] for x in host) / max(len(host), 1) * 1e3, 2) for i in range(4)] if host else None
    print(f"{kind:44} {dt:7.3f} ms/step   host ms [runner, zero_grad+backward, step+sched, consume]: {h}", flush=True)
    del model
    torch.cuda.empty_cache()


for kind in ("trainer", "trainer_mixup", "loop", "loop_mixup", "loop_mixup_sched", "loop_mixup_sched_metrics", "loop_mixup_sched_metrics_consume"):
    run(kind)

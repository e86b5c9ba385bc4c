"""Which weights should be updated inside their own weight-gradient GEMM epilogue?  Interleaved A/B in ONE process (guide rule 24; boxes of
this pool differ by > 10 % in step time, and separate processes on one box by several %): one Trainer per threshold
(runtime.fuse_min_elems: weights with fewer elements are left to the per-bucket update kernel), every round steps every trainer
`STEPS` times back to back, the table shows the median and the best round per threshold.

    python tools/fuse_ab.py [config] [batch] [threshold ...]        defaults: cfg2 64 0 4500000 13000000 1000000000
env: ROUNDS (6), STEPS (12)"""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import afft_amd  # noqa: E402
from afft_amd import dropout as D_, runtime as rt  # noqa: E402
from afft_amd.parallel import Trainer  # noqa: E402

ROUNDS, STEPS = int(os.environ.get("ROUNDS", 6)), int(os.environ.get("STEPS", 12))


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    ths = [int(x) for x in sys.argv[3:]] or [0, 4500000, 13000000, 1000000000]
    dev = torch.device("cuda", 0)
    afft_amd.set_precision("bf16")
    afft_amd.set_grad_mode("sink")
    wts = {"cls_action": 1.0, "past_cls_action": 1.0, "past_reg": 1.0}
    trainers = []
    feats = tgt = sub = None
    for th in ths:
        rt.set_fuse_min_elems(th)
        D_.manual_seed(42)
        model, c = bench.build_model(cfg, dev)
        model.train(True)
        if feats is None:
            feats, tgt, sub = bench.make_inputs(c, B, c["T"], 0, dev)
        tr = Trainer(model, wts)
        for _ in range(4):      # learns the bucket counts and the fused set under THIS threshold
            tr.step(feats, tgt, sub)
        torch.cuda.synchronize()
        trainers.append((th, tr, len(tr._fused or {})))
    times = {th: [] for th, _, _ in trainers}
    for _ in range(ROUNDS):
        for th, tr, _ in trainers:
            tr.step(feats, tgt, sub)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(STEPS):
                tr.step(feats, tgt, sub)
            torch.cuda.synchronize()
            times[th].append((time.perf_counter() - t0) / STEPS * 1e3)
    print(f"{cfg} B={B}: ms/step per threshold (median / best of {ROUNDS} rounds x {STEPS} steps), weights updated in epilogues")
    for th, _, n in trainers:
        t = times[th]
        print(f"  fuse_min_elems {th:>11d}: {statistics.median(t):7.3f} / {min(t):7.3f}   fused weights {n:3d}   rounds " + " ".join(f"{x:.2f}" for x in t))


if __name__ == "__main__":
    main()

"""Which GEMM sites of the 'fp16x2' forward can read their activation as ONE fp16 plane (runtime.one_pass_sites) inside the 1e-3 logits
tolerance, and what each buys.  One process, one model per configuration: for every site set the relative L2 error of the logits against
the exact-fp32 mode on `--clips` clips of `--seeds` different synthetic inputs (eval mode), the evaluation forward of the full batch, and
-- for the sets named by --train -- the training step (Trainer, fused update), interleaved with the all-sites-two-pass step in the same process.

usage (GPU box): python tools/lo_pass_sweep.py [--configs cfg2,ek100] [--train "conv1d;linear.fc2,conv1d"]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import afft_amd  # noqa: E402
from afft_amd import runtime as rt  # noqa: E402
import bench  # noqa: E402


def forward_logits(model, feats, tgt, sub):
    with torch.no_grad():
        o, _ = model(feats, mixup_fn=None, target=tgt, target_subclips=sub, target_subclips_ignore_index=None)
    return o["logits/action"]["all-fused"].double()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="cfg2,ek100")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--clips", type=int, default=8)
    ap.add_argument("--seeds", type=int, default=3)
    ap.add_argument("--sets", default=None, help="';'-separated site sets (each a ','-list); default: none, every site alone, linear, conv1d, all")
    ap.add_argument("--train", default="", help="';'-separated site sets whose training step is timed")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    sets = a.sets.split(";") if a.sets else ([""] + list(rt.ONE_PASS_SITES) + ["conv1d", "conv1d,linear.fc2", "conv1d,linear.fc2,linear.attn", "conv1d,linear.fc2,linear.attn,linear.proj",
                                                    "conv1d,linear.fc1,linear.fc2", "linear", "linear,conv1d"])
    for cfg_name in a.configs.split(","):
        print(f"== {cfg_name}: logits rel-L2 vs the exact-fp32 mode ({a.clips} clips x {a.seeds} inputs, max | mean), eval forward of {a.batch} clips", flush=True)
        inputs = []
        from afft_amd.config import BASELINE_CONFIGS
        c = BASELINE_CONFIGS[cfg_name]
        for seed in range(a.seeds):
            inputs.append(bench.make_inputs(c, a.clips, c["T"], 100 + seed, dev))
        afft_amd.set_precision("fp32")
        model, c = bench.build_model(cfg_name, dev)
        model.eval()
        refs = [forward_logits(model, *inp) for inp in inputs]
        afft_amd.set_precision("fp16x2")
        full = bench.make_inputs(c, a.batch, c["T"], 0, dev)
        for sites in sets:
            rt.set_one_pass_sites(sites)
            errs = [float(((forward_logits(model, *inp) - ref).norm() / ref.norm()).cpu()) for inp, ref in zip(inputs, refs)]
            for _ in range(2):
                forward_logits(model, *full)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(8):
                forward_logits(model, *full)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 8 * 1e3
            print(f"  {sites or '(none: every site two passes)':44s} err max {max(errs):.2e} mean {sum(errs) / len(errs):.2e}   forward {ms:6.2f} ms", flush=True)
        del model
        torch.cuda.empty_cache()
        train_sets = [s for s in a.train.split(";") if s.strip()]
        if train_sets:
            from afft_amd.parallel import Trainer
            wts = {"cls_action": 1.0, "past_cls_action": 1.0, "past_reg": 1.0}
            rt.set_one_pass_sites("")
            model, c = bench.build_model(cfg_name, dev)
            model.train()
            tr = Trainer(model, wts, bucket_elems=32 * 1024 * 1024)
            feats, tgt, sub = full
            for _ in range(5):
                tr.step(feats, tgt, sub)
            order = [""] + train_sets
            times = {s: [] for s in order}
            for rep in range(4):
                for s in order:
                    rt.set_one_pass_sites(s)
                    for _ in range(2):
                        tr.step(feats, tgt, sub)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(10):
                        tr.step(feats, tgt, sub)
                    torch.cuda.synchronize()
                    times[s].append((time.perf_counter() - t0) / 10 * 1e3)
            loss = float(tr.step(feats, tgt, sub)[0])
            for s in order:
                t = sorted(times[s])
                print(f"  train step, {s or '(none)':40s} {t[0]:6.2f} .. {t[-1]:6.2f} ms  (median {t[len(t) // 2]:6.2f}, {a.batch / t[len(t) // 2] * 1e3:7.1f} clips/s)", flush=True)
            print(f"  final loss {loss:.4f}", flush=True)
            del tr, model
            torch.cuda.empty_cache()
    rt.set_one_pass_sites("")


if __name__ == "__main__":
    import gc
    gc.collect()
    gc.freeze()
    main()

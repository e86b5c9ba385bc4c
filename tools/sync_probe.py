"""Which statement of Runner's metrics block makes the host wait for the GPU: each one is run behind ~30 ms of queued GPU work
and its host time is printed (a statement that returns in microseconds did not wait)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afft_amd.common.runner import LazyHostArray, accuracy  # noqa: E402

dev = "cuda:0"
a = torch.randn(8192, 8192, device=dev)
logits = torch.randn(64, 1, 3806, device=dev)
tgt = torch.rand(64, 3806, device=dev)


def busy():
    for _ in range(12):
        torch.mm(a, a)


def probe(name, fn):
    for rep in range(3):
        torch.cuda.synchronize()
        busy()
        t0 = time.perf_counter()
        r = fn()
        dt = (time.perf_counter() - t0) * 1e3
        torch.cuda.synchronize()
    print(f"{name:40} host {dt:8.3f} ms", flush=True)
    return r


_v, inds = probe("topk(tgt, 2)", lambda: torch.topk(tgt, 2, dim=1, largest=True, sorted=True))
rows = probe("arange", lambda: torch.arange(64, device=dev))
seq = probe("full_like", lambda: torch.full_like(rows, 0))
preds = probe("clone", lambda: logits.detach().clone())


def idx1():
    preds[rows, seq, inds[:, 0]] += preds[rows, seq, inds[:, 1]]


def idx2():
    preds[rows, seq, inds[:, 1]] = 0.0


probe("preds[idx] += preds[idx2]", idx1)
probe("preds[idx2] = 0", idx2)
probe("pinned empty", lambda: torch.empty(64, 3806, dtype=torch.float32, pin_memory=True))
probe("LazyHostArray(preds)", lambda: LazyHostArray(preds[:, 0, :].contiguous()))
probe("LazyHostArray(labels)", lambda: LazyHostArray(inds[:, 0].contiguous()))
probe("accuracy", lambda: accuracy(preds, inds[:, :1], topk=(1, 5)))

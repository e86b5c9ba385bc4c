"""Round 6: one enqueue of ~60-95 ms among the first ~30 training steps of a process (host_gpu_split.py: "host enqueue p50 7.6 max 64"), which
the GPU feels when it falls inside bench.py's short timed region right after a synchronisation (the host's lead is still small): is it
Python's cyclic garbage collector walking the whole heap?  Logs every collection (generation, ms) and the slowest enqueues with their step
index over 60 steps, with and without gc.freeze() after set-up.   usage: python tools/gc_probe.py [freeze]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
import afft_amd
from afft_amd.parallel import Trainer
freeze = len(sys.argv) > 1 and sys.argv[1] == "freeze"
afft_amd.set_precision("bf16")
dev = torch.device("cuda:0")
model, c = B.build_model("cfg2", dev)
feats, tgt, sub = B.make_inputs(c, 64, c["T"], 0, dev)
tr = Trainer(model, {"cls_action": 1.0, "past_cls_action": 1.0, "past_reg": 1.0})
model.train()
events, t_gc = [], [0.0]
def cb(phase, info):
    if phase == "start":
        t_gc[0] = time.perf_counter()
    else:
        events.append((info["generation"], (time.perf_counter() - t_gc[0]) * 1e3, info["collected"]))
gc.callbacks.append(cb)
if freeze:
    gc.collect()
    gc.freeze()
host = []
for i in range(5):
    tr.step(feats, tgt, sub)
torch.cuda.synchronize()
t0 = time.perf_counter()
marks = []
for i in range(60):
    n_ev = len(events)
    h0 = time.perf_counter()
    tr.step(feats, tgt, sub)
    host.append(((time.perf_counter() - h0) * 1e3, i, [e for e in events[n_ev:]]))
    if i == 19:
        torch.cuda.synchronize()
        marks.append((time.perf_counter() - t0) / 20 * 1e3)
        t1 = time.perf_counter()
torch.cuda.synchronize()
marks.append((time.perf_counter() - t1) / 40 * 1e3)
slow = sorted(host, reverse=True)[:4]
print(f"{'freeze' if freeze else 'plain '}: first 20 steps {marks[0]:.2f} ms/step, next 40 {marks[1]:.2f} | slowest enqueues (ms, step, gc events in it): " +
      "; ".join(f"{h:.1f} @{i} {[(g, round(ms, 1)) for g, ms, _ in ev]}" for h, i, ev in slow) + f" | gen2 collections {[round(ms, 1) for g, ms, _ in events if g == 2]} objects {len(gc.get_objects())}")

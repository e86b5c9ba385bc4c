"""Is a process's step bound by the host or by the GPU?  Per process: the CPUs it may run on, the CPU it runs on, the load average, the
host's enqueue time per training step (p50 / max) and the whole step time (enqueue + drain; 30 steps after 8 warm-up steps, as bench.py
times them).  Round 6: sequential bench.py processes on ONE box gave 11.7 .. 14.7 ms/step for the same build.
usage: python tools/host_gpu_split.py [config] [batch] [tag]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
import afft_amd
from afft_amd.parallel import Trainer
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
tag = sys.argv[3] if len(sys.argv) > 3 else ""
afft_amd.set_precision("bf16")
dev = torch.device("cuda:0")
if os.environ.get("PREALLOC_FIRST_GB"):      # ... before the model itself exists
    x = torch.empty(int(os.environ["PREALLOC_FIRST_GB"]) << 30, dtype=torch.uint8, device=dev)
    del x
model, c = B.build_model(name, dev)
feats, tgt, sub = B.make_inputs(c, batch, c["T"], 0, dev)
if os.environ.get("PREALLOC_GB"):      # one big block first: later allocations are carved out of ONE hipMalloc'ed segment
    x = torch.empty(int(os.environ["PREALLOC_GB"]) << 30, dtype=torch.uint8, device=dev)
    del x
tr = Trainer(model, {"cls_action": 1.0, "past_cls_action": 1.0, "past_reg": 1.0})
model.train()
for _ in range(8):
    tr.step(feats, tgt, sub)
torch.cuda.synchronize()
import threading
probe, _src = B._power_probe()
samples, stop = [], threading.Event()
def poll():
    while not stop.is_set():
        try:
            samples.append(probe())
        except Exception:
            pass
        time.sleep(0.02)
th = threading.Thread(target=poll)
if probe is not None:
    th.start()
host = []
t_all = time.perf_counter()
for _ in range(30):
    t0 = time.perf_counter()
    tr.step(feats, tgt, sub)
    host.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
step = (time.perf_counter() - t_all) / 30 * 1e3
stop.set()
if probe is not None:
    th.join()
pw = sum(x[0] for x in samples) / max(1, len(samples))
ck = sum(x[1] for x in samples) / max(1, len(samples))
# the same 30 steps with the host waiting for the GPU after every step: the GPU's own time per step (nothing queued behind it)
g = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.step(feats, tgt, sub)
    torch.cuda.synchronize()
    g.append((time.perf_counter() - t0) * 1e3)
# the same step with the weight gradients on the MAIN stream (nothing overlaps the backward chain)
from afft_amd import runtime as rt
rt.set_overlap_wgrad(False)
for _ in range(4):
    tr.step(feats, tgt, sub)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    tr.step(feats, tgt, sub)
torch.cuda.synchronize()
serial = (time.perf_counter() - t0) / 20 * 1e3
rt.set_overlap_wgrad(True)
for _ in range(4):
    tr.step(feats, tgt, sub)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    tr.step(feats, tgt, sub)
torch.cuda.synchronize()
again = (time.perf_counter() - t0) / 20 * 1e3
hs = sorted(host)
aff = os.sched_getaffinity(0)
print(f"{tag} {name} B={batch}: step {step:.2f} ms, weight gradients on the main stream {serial:.2f}, two streams again {again:.2f} | host enqueue p50 {hs[15]:.2f} max {hs[-1]:.2f} | synced step p50 {sorted(g)[5]:.2f} | "
      f"power {pw:.0f} W sclk {ck:.0f} MHz ({len(samples)} samples) | mem reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB | cpus allowed {len(aff)} on cpu {open('/proc/self/stat').read().split()[38]} load {os.getloadavg()[0]:.1f} threads {torch.get_num_threads()}")

"""Is the device's HBM throughput steady?  (round 6: the training step runs in a "fast" or a "slow" state, in which HBM-bound kernels take 2x.)
A framework-free probe: a 1-GiB device-to-device copy (torch, 2 GiB of traffic) and a 8192^3 bf16 GEMM (torch.mm: the vendor library), timed
with events in windows of ~0.25 s for `seconds`; prints TB/s, TFLOP/s, package power and sclk per window and a histogram at the end.
usage: python tools/hbm_probe.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
dev = torch.device("cuda:0")
src = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_()
dst = torch.empty_like(src)
a = torch.randn(8192, 8192, device=dev).to(torch.bfloat16)
b = torch.randn(8192, 8192, device=dev).to(torch.bfloat16)
probe, _ = B._power_probe()
ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
t_end = time.perf_counter() + seconds
rows = []
while time.perf_counter() < t_end:
    e0, e1, e2 = ev(), ev(), ev()
    e0.record()
    for _ in range(40):
        dst.copy_(src)
    e1.record()
    for _ in range(8):
        torch.mm(a, b)
    e2.record()
    torch.cuda.synchronize()
    tbs = 40 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e12
    tf = 8 * 2 * 8192 ** 3 / (e1.elapsed_time(e2) * 1e-3) / 1e12
    pw, ck = probe() if probe else (0, 0)
    rows.append((tbs, tf))
    print(f"t={seconds - (t_end - time.perf_counter()):5.1f}s copy {tbs:5.2f} TB/s  gemm {tf:7.1f} TF/s  {pw:6.0f} W {ck} MHz", flush=True)
cs = sorted(r[0] for r in rows)
gs = sorted(r[1] for r in rows)
print(f"copy TB/s min {cs[0]:.2f} p10 {cs[len(cs) // 10]:.2f} median {cs[len(cs) // 2]:.2f} max {cs[-1]:.2f} | gemm TF/s min {gs[0]:.0f} median {gs[len(gs) // 2]:.0f} max {gs[-1]:.0f} | windows {len(rows)}")

"""TFLOP/s per W of the 256x256 kernels on 8192^3 (VERDICT r4 #3's criterion): the GEMM loops for 4 s while package power and the
graphics clock are polled IN PROCESS (bench._power_probe: amdgpu hwmon of this GPU's PCI address).  usage: python tools/q4_power.py"""
import os, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import bench as B
from afft_amd import _lib, ops

dev = "cuda:0"
M = N = K = 8192
g = torch.Generator().manual_seed(0)
a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
b = torch.randn(N, K, generator=g).to(torch.bfloat16).to(dev)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
read, src = B._power_probe()
print("power source:", src)


def run(name, fn, seconds=4.0):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    stop, samples = threading.Event(), []

    def poll():
        while not stop.is_set():
            try:
                samples.append(read())
            except Exception:
                pass
            time.sleep(0.05)
    th = threading.Thread(target=poll)
    th.start()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(10):
            fn()
        n += 10
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    stop.set(); th.join()
    pw = sorted(s[0] for s in samples[4:]); ck = sorted(s[1] for s in samples[4:])
    tf = 2.0 * M * N * K / dt / 1e12
    w = sum(pw) / len(pw)
    print(f"{name:28} {dt * 1e3:7.3f} ms {tf:7.1f} TFLOP/s  {w:7.1f} W avg  {ck[len(ck) // 2]:5d} MHz  {tf / w:5.3f} TFLOP/s per W")


for v, name in ((3, "ping-pong (variant 3)"), (11, "four-quadrant (variant 11)")):
    def f(v=v):
        _lib.check(_lib.lib().afft_set_gemm_variant(v))
        ops.gemm(a, b, out, b_t=True)
    run(name, f)
_lib.check(_lib.lib().afft_set_gemm_variant(0))
bt = b.t()
run("vendor (torch.mm)", lambda: torch.mm(a, bt, out=out))

"""A/B of the 128x128 GEMM kernel on v_mfma_f32_16x16x32_bf16 (product) against an experimental build on v_mfma_f32_32x32x16_bf16
(-DAFFT_G128_MFMA32=1), both loaded in one process: results, time, clock and power (rocm-smi).
usage: python tools/m32_ab.py base.so m32.so"""
import ctypes as C
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bd_ab import desc, load  # noqa: E402
from tools.power_probe import smi  # noqa: E402

DEV = "cuda:0"


def loop(lib, d, st, seconds):
    stop, samples = threading.Event(), []

    def poll():
        while not stop.is_set():
            samples.append(smi())
            time.sleep(0.05)
    for _ in range(5):
        lib.afft_gemm(C.byref(d), st)
    torch.cuda.synchronize()
    th = threading.Thread(target=poll)
    th.start()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(8):
            lib.afft_gemm(C.byref(d), st)
        n += 8
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    stop.set()
    th.join()
    pw = [s[0] for s in samples if s[0] is not None]
    clk = sorted(int(s[1].strip("()Mhz")) for s in samples if s[1] and s[1].startswith("("))
    return dt, sum(pw) / max(1, len(pw)), clk[len(clk) // 2] if clk else 0


def main():
    libs = [load(p) for p in sys.argv[1:3]]
    names = ["16x16x32", "32x32x16"]
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(0)
    for (M, N, K) in ((8192, 8192, 8192), (5120, 2048, 2048), (1024, 8192, 2048), (1024, 2048, 2048), (1088, 3840, 2048)):
        a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(DEV)
        b = torch.randn(N, K, generator=g).to(torch.bfloat16).to(DEV)
        outs = [torch.zeros(M, N, dtype=torch.bfloat16, device=DEV) for _ in libs]
        row = []
        for lib, o, nm in zip(libs, outs, names):
            lib.afft_set_gemm_variant(1)
            lib.afft_set_gemm_splitk.argtypes = [C.c_int]
            lib.afft_set_gemm_splitk(0)
            d = desc(a, b, o)
            assert lib.afft_gemm(C.byref(d), st) == 0
            torch.cuda.synchronize()
            dt, pw, clk = loop(lib, d, st, 4.0 if M * N * K > 1e11 else 2.0)
            tf = 2.0 * M * N * K / dt / 1e12
            row.append(f"{nm}: {dt * 1e6:8.1f} us {tf:6.0f} TF  {pw:6.0f} W  {clk} MHz  {tf / max(clk, 1) * 1e3:5.0f} TF/GHz")
        ref = (a.float() @ b.float().t())
        errs = [float((o.float() - ref).norm() / ref.norm()) for o in outs]
        print(f"{M}x{N}x{K}: " + " | ".join(row) + f" | rel err vs fp32 torch {errs[0]:.2e} {errs[1]:.2e}", flush=True)


if __name__ == "__main__":
    main()

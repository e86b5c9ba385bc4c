"""Clock and power of the GPU while a workload runs (rocm-smi polled from a thread): is the part power-limited under the step, under
our GEMM kernels and under the vendor's?  usage: python tools/power_probe.py"""
import json
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afft_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout
        j = json.loads(out[out.index("{"):])
        c = next(iter(j.values()))
        pw = next((float(v) for k, v in c.items() if "Power" in k and "(W)" in k), None)
        sclk = next((v for k, v in c.items() if k.startswith("sclk")), None)
        mclk = next((v for k, v in c.items() if k.startswith("mclk")), None)
        return pw, sclk, mclk
    except Exception as ex:  # noqa: BLE001
        return None, repr(ex)[:80], None


def probe(name, fn, seconds=6.0):
    stop, samples = threading.Event(), []

    def poll():
        while not stop.is_set():
            samples.append(smi())
            time.sleep(0.05)
    fn()
    torch.cuda.synchronize()
    th = threading.Thread(target=poll)
    th.start()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        fn()
        n += 1
        if n % 4 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    pws = [s[0] for s in samples if s[0] is not None]
    clk = [s[1] for s in samples]
    print(f"{name:46} {dt / n * 1e3:9.3f} ms/iter  power avg {sum(pws) / max(1, len(pws)):6.1f} W max {max(pws) if pws else 0:6.1f} W  "
          f"sclk samples {sorted(set(clk))[:6]}  mclk {sorted(set(s[2] for s in samples))[:3]}  ({len(samples)} samples)", flush=True)
    return dt / n


def main():
    print("idle:", smi())
    g = torch.Generator().manual_seed(0)
    M = N = K = 8192
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    b = torch.randn(N, K, generator=g).to(torch.bfloat16).to(dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    fl = 2.0 * M * N * K
    t = probe("vendor torch.mm 8192^3 (yard-stick)", lambda: torch.mm(a, b.t(), out=out))
    print(f"   -> {fl / t / 1e12:.0f} TFLOP/s")
    _lib.check(_lib.lib().afft_set_gemm_variant(3))
    t = probe("ping-pong 256x256 kernel 8192^3", lambda: ops.gemm(a, b, out, b_t=True))
    print(f"   -> {fl / t / 1e12:.0f} TFLOP/s")
    pk = torch.empty(N * K, dtype=torch.bfloat16, device=dev)
    ops.pack_weight(b.float(), pk)
    _lib.check(_lib.lib().afft_set_gemm_variant(9))
    t = probe("B-direct 256x256 kernel 8192^3 (packed B)", lambda: ops.gemm(a, pk.view(N, K), out, b_t=True))
    print(f"   -> {fl / t / 1e12:.0f} TFLOP/s")
    _lib.check(_lib.lib().afft_set_gemm_variant(0))
    big = torch.zeros(256 << 20, dtype=torch.float32, device=dev)
    big2 = torch.zeros(256 << 20, dtype=torch.float32, device=dev)
    t = probe("copy 1 GiB (HBM-bound)", lambda: big2.copy_(big))
    print(f"   -> {2 * big.numel() * 4 / t / 1e12:.2f} TB/s")
    del big, big2
    import bench as B
    import afft_amd
    from afft_amd.parallel import Trainer
    afft_amd.set_precision("bf16")
    model, c = B.build_model("cfg2", dev)
    feats, tgt, sub = B.make_inputs(c, 64, c["T"], 0, dev)
    tr = Trainer(model, {"cls_action": 1.0, "past_cls_action": 1.0, "past_reg": 1.0})
    model.train()
    for _ in range(5):
        tr.step(feats, tgt, sub)
    probe("training step cfg2 B = 64", lambda: tr.step(feats, tgt, sub), seconds=8.0)
    model.eval()
    with torch.no_grad():
        probe("evaluation forward cfg2 B = 64", lambda: model(feats, mixup_fn=None, target=tgt, target_subclips=sub, target_subclips_ignore_index=None))


if __name__ == "__main__":
    main()

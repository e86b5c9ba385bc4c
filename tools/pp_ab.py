"""Interleaved A/B of several builds of the library's 256x256 bf16 GEMM in ONE process (guide rule 24): every library is loaded with
its own ctypes handle, every round times every (library, shape) pair back to back, the table shows median and best per pair.
Operands are uniform random in [-1, 1) (rule 25).  Before timing, every library's result is checked against an fp32 torch product
(relative L2) on every layout, and each timed shape is run `RACE` times and compared BITWISE with its first result (a race in the
LDS ring shows up as a run-to-run difference).

    python tools/pp_ab.py afft_amd/lib/libafft_hip.so afft_amd/lib/libafft_hip_x.so ...
env: ROUNDS (5), ITERS (20), RACE (10), SHAPES=path|big|all, VARIANT (3 = force the 256x256 kernels; 0 = the library's dispatch)"""
import ctypes as C
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afft_amd import _lib as L  # noqa: E402

dev = "cuda:0"
ROUNDS, ITERS, RACE = int(os.environ.get("ROUNDS", 5)), int(os.environ.get("ITERS", 20)), int(os.environ.get("RACE", 10))
VARIANT = int(os.environ.get("VARIANT", 3))
PATH = [("nt", 5120, 6144, 2048), ("nt", 5120, 8192, 2048), ("nt", 5120, 2048, 8192), ("nn", 5120, 2048, 6144), ("nn", 5120, 8192, 2048),
        ("nn", 5120, 2048, 8192), ("tn", 6144, 2048, 5120), ("tn", 8192, 2048, 5120), ("tn", 2048, 8192, 5120), ("tn", 8192, 2048, 1024)]
BIG = [("nt", 4096, 4096, 4096), ("nt", 8192, 8192, 8192)]
# the small-grid launches of the step (128x128 tiles, VARIANT=0 or 1): predictor M = B*T = 1024 rows, EK100 widths, weight gradients over 1024 rows
SMALL = [("nn", 1024, 6144, 2048), ("nn", 1024, 2048, 2048), ("nn", 1024, 8192, 2048), ("nn", 1024, 2048, 8192), ("nt", 1024, 6144, 2048),
         ("nt", 1024, 2048, 2048), ("nt", 1024, 8192, 2048), ("nt", 1024, 2048, 8192), ("tn", 2048, 2048, 1024), ("tn", 2048, 2048, 5120),
         ("nt", 5120, 1024, 1024), ("nn", 5120, 1024, 4096), ("tn", 1024, 3072, 5120), ("nt", 1280, 6144, 2048), ("tn", 2048, 6144, 1280)]
# the fuser's forward GEMMs in the fp16x2 precision: fp16 hi segment + block-scaled fp8 lo segment (afft_gemm_t.split3 = 3), NT
LO8 = [("lo8", 5120, 6144, 2048), ("lo8", 5120, 2048, 2048), ("lo8", 5120, 8192, 2048), ("lo8", 5120, 2048, 8192), ("lo8", 1024, 2048, 2048)]
# one fp16 pass (afft_gemm_t.split3 = 4, F16=1 is implied): the predictor's GEMMs of the fp16x2 forward (NN, M = B*T rows) and the fusers' fc2
H1 = [("nn", 1024, 6144, 2048), ("nn", 1024, 2048, 2048), ("nn", 1024, 8192, 2048), ("nn", 1024, 2048, 8192), ("nt", 5120, 2048, 8192),
      ("nn", 1024, 3072, 1024), ("nn", 1024, 1024, 1024), ("nn", 1024, 2048, 1024), ("nt", 5120, 1024, 4096)]
SHAPES = {"path": PATH, "big": BIG, "all": PATH + BIG, "small": SMALL, "lo8": LO8, "h1": H1}[os.environ.get("SHAPES", "all")]
F16 = os.environ.get("F16", "0") == "1" or os.environ.get("SHAPES") == "h1"      # fp16 operands, one pass


def load(path):
    lib = C.CDLL(os.path.abspath(path))
    lib.afft_last_error.restype = C.c_char_p
    for name in ("afft_gemm", "afft_set_gemm_variant", "afft_set_gemm_splitk"):
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = L._SIGS[name]
    assert lib.afft_set_gemm_variant(VARIANT) == 0
    return lib


def operands(layout, M, N, K, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    if layout == "lo8":      # (hi fp16 plane, e4m3(2^11 (a - hi))) of the activation, (fp16 image, e4m3(2^8 w)) of the weight: what the producers write
        from afft_amd import ops
        a32 = (torch.rand(M, K, generator=g) * 2 - 1).to(dev)
        w32 = ((torch.rand(N, K, generator=g) * 2 - 1) * 0.05).to(dev)
        hi = torch.empty(M, K, dtype=torch.float16, device=dev)
        a8 = torch.empty(M, K, dtype=torch.uint8, device=dev)
        ops.quant_e4m3(a32, 2048.0, a8, hi=hi)
        w8 = torch.empty(N, K, dtype=torch.uint8, device=dev)
        ops.quant_e4m3(w32, 256.0, w8)
        return (hi, a8, a32), (w32.half(), w8, w32)
    u = lambda *s: (torch.rand(*s, generator=g) * 2 - 1).to(torch.float16 if F16 else torch.bfloat16).to(dev)      # noqa: E731
    if layout == "nt":
        return u(M, K), u(N, K)
    if layout == "nn":
        return u(M, K), u(K, N)
    return u(K, M), u(K, N)


def desc(layout, a, b, out):
    d = L.GemmDesc()
    if layout == "lo8":
        (hi, a8, _), (w16, w8, _) = a, b
        M, K = hi.shape
        N = w16.shape[0]
        d.M, d.N, d.K, d.dtype = M, N, K, L.BF16
        d.A, d.a_rs, d.a_cs = hi.data_ptr(), hi.stride(0), 1
        d.B, d.b_rs, d.b_cs = w16.data_ptr(), 1, w16.stride(0)
        d.split3 = 3
        d.a8, d.a8_ld, d.b8, d.b8_ld = a8.data_ptr(), a8.stride(0), w8.data_ptr(), w8.stride(0)
        d.alpha = 1.0
        d.out, d.ldo, d.out_dtype = out.data_ptr(), out.stride(0), L.F32
        d.workspace, d.workspace_bytes = WS.data_ptr(), WS.numel()
        return d
    a_t, b_t = layout == "tn", layout == "nt"
    M, K = (a.shape[1], a.shape[0]) if a_t else a.shape
    N = b.shape[0] if b_t else b.shape[1]
    d.M, d.N, d.K, d.dtype = M, N, K, L.BF16
    d.A, d.B = a.data_ptr(), b.data_ptr()
    d.a_rs, d.a_cs = (a.stride(1), a.stride(0)) if a_t else (a.stride(0), a.stride(1))
    d.b_rs, d.b_cs = (b.stride(1), b.stride(0)) if b_t else (b.stride(0), b.stride(1))
    d.alpha = 1.0
    if F16:
        d.split3 = 4
    d.out, d.ldo, d.out_dtype = out.data_ptr(), out.stride(0), (L.F32 if out.dtype == torch.float32 else L.BF16)
    d.workspace, d.workspace_bytes = WS.data_ptr(), WS.numel()      # split-K scratch (counters zero between launches)
    return d


WS = torch.zeros(192 << 20, dtype=torch.uint8, device=dev)


def run(lib, d, n=1):
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(n):
        rc = lib.afft_gemm(C.byref(d), s)
        if rc:
            raise RuntimeError(lib.afft_last_error().decode())


def reference(layout, a, b):
    if layout == "lo8":
        return a[2].double() @ b[1 - 1].double().t() if False else (a[2].double() @ b[0].double().t()).float()      # exact activation x fp16-rounded weight
    a32, b32 = a.float(), b.float()
    return (a32.t() if layout == "tn" else a32) @ (b32.t() if layout == "nt" else b32)


def main():
    paths = sys.argv[1:] or [L.LIB_PATH]
    libs = [(os.path.basename(p).replace("libafft_hip", "").replace(".so", "") or "product", load(p)) for p in paths]
    print("libraries:", ", ".join(n for n, _ in libs), f"| variant {VARIANT} rounds {ROUNDS} iters {ITERS} race repeats {RACE}")
    # correctness: every layout, a shape with several tiles and K-tile pairs, plus one edge shape that must take the general kernel
    checks = (("nt", 512, 768, 1024), ("nn", 512, 768, 1024), ("tn", 768, 512, 1024), ("nt", 1088, 3840, 2048), ("tn", 512, 512, 320),
              ("nt", 256, 256, 256), ("tn", 2048, 2048, 1024), ("nn", 1024, 2048, 8192), ("nt", 1024, 2048, 8192), ("tn", 1024, 1024, 1280),
              ("nn", 384, 640, 896))
    if F16:
        checks = tuple(c for c in checks if c[0] != "tn")      # forward layouts only
    if os.environ.get("SHAPES") == "lo8":      # whole tiles (steady-state kernel), edge tiles and a K-tile count that is not a multiple of 4 (general kernel)
        checks = (("lo8", 512, 768, 1024), ("lo8", 2560, 2048, 256), ("lo8", 2560, 2304, 2048), ("lo8", 2500, 2048, 1024), ("lo8", 2560, 2048, 384))
    for layout, M, N, K in checks:
        a, b = operands(layout, M, N, K, seed=1)
        ref = reference(layout, a, b)
        for name, lib in libs:
            out = torch.zeros(M, N, dtype=torch.float32 if layout in ("tn", "lo8") else torch.bfloat16, device=dev)
            run(lib, desc(layout, a, b, out))
            torch.cuda.synchronize()
            err = float((out.float() - ref).norm() / ref.norm())
            flag = "" if err < (1e-5 if layout == "tn" else 2e-4 if layout == "lo8" else 4e-3) else "   <-- WRONG"
            print(f"check {layout} {M}x{N}x{K} {name:>10}: rel err {err:.2e}{flag}")
    times = {(n, s): [] for n, _ in libs for s in SHAPES}
    races = {}
    data = {}
    for s in SHAPES:
        layout, M, N, K = s
        a, b = operands(layout, M, N, K)
        data[s] = (a, b, {n: torch.zeros(M, N, dtype=torch.float32 if layout in ("tn", "lo8") else torch.bfloat16, device=dev) for n, _ in libs})
    for (layout, M, N, K) in SHAPES:      # race screen
        a, b, outs = data[(layout, M, N, K)]
        for name, lib in libs:
            d = desc(layout, a, b, outs[name])
            run(lib, d)
            first = outs[name].clone()
            bad = 0
            for _ in range(RACE):
                run(lib, d)
                bad += int(not torch.equal(outs[name], first))
            races[(name, (layout, M, N, K))] = bad
    ev = lambda: torch.cuda.Event(enable_timing=True)      # noqa: E731
    for _ in range(ROUNDS):
        for s in SHAPES:
            a, b, outs = data[s]
            for name, lib in libs:
                d = desc(s[0], a, b, outs[name])
                run(lib, d, 3)
                e0, e1 = ev(), ev()
                e0.record()
                run(lib, d, ITERS)
                e1.record()
                torch.cuda.synchronize()
                times[(name, s)].append(e0.elapsed_time(e1) / ITERS)
    print(f"{'layout':6} {'M':>5} {'N':>5} {'K':>5} | " + " | ".join(f"{n:>10} med us / TF (best TF)" for n, _ in libs) + " | ratio of medians to the first")
    for s in SHAPES:
        layout, M, N, K = s
        fl = 2.0 * M * N * K
        cells, meds = [], []
        for name, _ in libs:
            t = times[(name, s)]
            med, best = statistics.median(t), min(t)
            meds.append(med)
            r = races[(name, s)]
            cells.append(f"{med * 1e3:8.1f} {fl / med / 1e9:7.1f} ({fl / best / 1e9:7.1f}){' RACE ' + str(r) if r else ''}")
        print(f"{layout:6} {M:5d} {N:5d} {K:5d} | " + " | ".join(cells) + " | " + " ".join(f"{meds[0] / m:5.3f}" for m in meds))


if __name__ == "__main__":
    main()

"""A/B of library builds in ONE process (cdna guide rule 24): every .so given on the command line is loaded side by side and the
NT GEMM shapes below are timed on each, interleaved, over several rounds.  Usage:
    python tools/bd_ab.py VARIANT[,VARIANT..] lib1.so lib2.so ...      (VARIANT = afft_set_gemm_variant code, e.g. 8)
Diagnostic builds (AFFT_BD_DIAG) compute wrong results: timing only."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afft_amd import _lib  # noqa: E402

DEV = "cuda:0"
SHAPES = [(5120, 2048, 8192), (5120, 2048, 2048), (5120, 6144, 2048), (5120, 8192, 2048), (8320, 8192, 8192)]
if os.environ.get("SHAPES"):
    SHAPES = [tuple(int(x) for x in s.split("x")) for s in os.environ["SHAPES"].split(",")]


def load(path):
    lib = C.CDLL(path)
    lib.afft_gemm.argtypes = [C.POINTER(_lib.GemmDesc), C.c_void_p]
    lib.afft_gemm.restype = C.c_int
    lib.afft_set_gemm_variant.argtypes = [C.c_int]
    lib.afft_last_error.restype = C.c_char_p
    return lib


LAYOUT = os.environ.get("LAYOUT", "NT")      # NT: A [M,K], B [N,K] (forward) | NN: B [K,N] (data gradient) | TN: A [K,M], B [K,N] (weight gradient)


def desc(a, b, out):
    d = _lib.GemmDesc()
    M, N = out.shape
    K = a.shape[0] if LAYOUT == "TN" else a.shape[1]
    d.M, d.N, d.K, d.dtype = M, N, K, _lib.BF16
    d.A = a.data_ptr()
    d.a_rs, d.a_cs = (1, a.stride(0)) if LAYOUT == "TN" else (a.stride(0), 1)
    d.B = b.data_ptr()
    d.b_rs, d.b_cs = (1, b.stride(0)) if LAYOUT == "NT" else (b.stride(0), 1)
    d.alpha = 1.0
    d.out, d.ldo, d.out_dtype = out.data_ptr(), out.stride(0), _lib.BF16
    return d


def main():
    variants = [int(v) for v in sys.argv[1].split(",")]
    paths = sys.argv[2:]
    libs = [load(p) for p in paths]
    names = [os.path.basename(p).replace("libafft_hip", "").replace(".so", "") or "base" for p in paths]
    g = torch.Generator().manual_seed(0)
    st = torch.cuda.current_stream().cuda_stream
    rounds = int(os.environ.get("ROUNDS", "3"))
    cold = os.environ.get("COLD") == "1"       # COLD=1: one launch at a time, a 1-GiB fill in front of it (weights and activations from HBM)
    flush = torch.empty(1 << 28, dtype=torch.float32, device=DEV) if cold else None
    for (M, N, K) in SHAPES:
        a = torch.randn(*((K, M) if LAYOUT == "TN" else (M, K)), generator=g).to(torch.bfloat16).to(DEV)
        b = torch.randn(*((N, K) if LAYOUT == "NT" else (K, N)), generator=g).to(torch.bfloat16).to(DEV)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        from tools.bd_check import pack_b
        bp = pack_b(b) if LAYOUT == "NT" else b
        d0, dp = desc(a, b, out), desc(a, bp, out)
        res = {}
        for rnd in range(rounds):
            for v in variants:
                for lib, nm in zip(libs, names):
                    lib.afft_set_gemm_variant(v)
                    d = dp if v >= 9 else d0
                    for _ in range(3):
                        rc = lib.afft_gemm(C.byref(d), st)
                        assert rc == 0, lib.afft_last_error()
                    torch.cuda.synchronize()
                    if cold:
                        ts = []
                        for _ in range(8):
                            flush.fill_(1.0)
                            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            torch.cuda.synchronize()
                            s.record()
                            lib.afft_gemm(C.byref(d), st)
                            e.record()
                            torch.cuda.synchronize()
                            ts.append(s.elapsed_time(e) * 1e3)
                        res.setdefault((v, nm), []).append(sorted(ts)[len(ts) // 2])
                        continue
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    for _ in range(20):
                        lib.afft_gemm(C.byref(d), st)
                    e.record()
                    torch.cuda.synchronize()
                    res.setdefault((v, nm), []).append(s.elapsed_time(e) / 20 * 1e3)
        fl = 2.0 * M * N * K
        print(f"{M}x{N}x{K}: " + " | ".join(f"v{v}{nm} {min(t):7.1f} us {fl / min(t) / 1e6:5.0f} TF" for (v, nm), t in res.items()), flush=True)


if __name__ == "__main__":
    main()

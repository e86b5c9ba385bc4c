"""Stream-K of the 256x256 kernel against the plain grid over K: the slope is the cost of a K-iteration, the intercept the fixed
cost (prologue, hand-over, epilogue).  usage: python tools/sk_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import bench  # noqa: E402

if __name__ == "__main__":
    print(f"{'layout':6} {'M':>6} {'N':>6} {'K':>6} | plain us | stream-K us | 128x128 us")
    for lay, M, N, Ks in (("nt", 5120, 2048, (1024, 2048, 4096, 8192, 16384)), ("nt", 5120, 8192, (512, 1024, 2048, 4096)),
                          ("tn", 6144, 2048, (1024, 2560, 5120)), ("nn", 5120, 2048, (2048, 6144, 8192)), ("tn", 2048, 2048, (1024, 5120)),
                          ("nt", 1024, 2048, (2048, 8192)), ("nt", 1024, 8192, (2048,))):
        for K in Ks:
            r = [bench(lay, M, N, K, v)[0] * 1e3 for v in (30, 33, 1)]
            print(f"{lay:6} {M:6d} {N:6d} {K:6d} | {r[0]:8.1f} | {r[1]:11.1f} | {r[2]:10.1f}")

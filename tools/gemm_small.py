"""Variant sweep on the small-M (GPT-2, M = B*T = 1024) and small-output GEMM shapes of the cfg2 step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import bench
shapes = [("nn", 1024, 6144, 2048), ("nn", 1024, 2048, 2048), ("nn", 1024, 8192, 2048), ("nn", 1024, 2048, 8192),
          ("nt", 1024, 2048, 6144), ("nt", 1024, 2048, 2048), ("nt", 1024, 2048, 8192), ("nt", 1024, 8192, 2048),
          ("tn", 2048, 6144, 1024), ("tn", 2048, 2048, 1024), ("tn", 2048, 8192, 1024), ("tn", 8192, 2048, 1024),
          ("tn", 2048, 2048, 5120), ("nt", 1088, 3840, 2048), ("nn", 1088, 2048, 3840), ("tn", 3840, 2048, 1088)]
vs = [int(v) for v in os.environ.get("VARIANTS", "1,4,5,3").split(",")]
print("layout M N K | " + " | ".join(f"v{v} TF" for v in vs))
for lay, M, N, K in shapes:
    r = [bench(lay, M, N, K, v, iters=20)[1] for v in vs]
    print(f"{lay} {M} {N} {K} | " + " | ".join(f"{x:7.1f}" for x in r))

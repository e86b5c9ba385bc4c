"""Times the ping-pong GEMM (variant 3) on a few NT shapes. Run with AFFT_LIB=<tuning build>."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import bench
shapes = [(8192, 8192, 8192), (5120, 6144, 2048), (5120, 8192, 2048), (5120, 2048, 8192)]
res = []
for M, N, K in shapes:
    ms, tf = bench("nt", M, N, K, 3, iters=20)
    res.append(f"{M}x{N}x{K}: {tf:7.1f}")
print(os.environ.get("AFFT_LIB", "default").split("/")[-1], " | ".join(res))

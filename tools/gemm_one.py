"""Runs one bf16 GEMM shape repeatedly (for rocprofv3 --pmc runs). usage: gemm_one.py layout M N K variant iters"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import bench
lay, M, N, K, var, it = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
print(bench(lay, M, N, K, var, it))

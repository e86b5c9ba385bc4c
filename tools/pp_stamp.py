"""Diagnostic: per-segment cycle shares of the ping-pong GEMM (needs `tools/lib_variant_one.sh stamp gemm_pp -DAFFT_DIAG_BUILD -DAFFT_PP_STAMP -DAFFT_PP2=0`:
into afft_amd/lib/libafft_hip_stamp.so). Prints, for each wave of workgroup 0, average cycles per phase spent in
L (LDS reads + DMA issue), W (vmcnt wait), B1 (barrier after L), C (16 MFMAs), B2 (barrier after C)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afft_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libafft_hip_stamp.so")
from afft_amd import ops
lib = _lib.lib()
lib.afft_debug_pp_stamp.argtypes = [ctypes.c_void_p]
buf = torch.zeros(64, dtype=torch.int64, device="cuda:0")
lib.afft_debug_pp_stamp(buf.data_ptr())
_lib.check(lib.afft_set_gemm_variant(3))
M, N, K = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (8192, 8192, 8192))]
a = torch.randn(M, K).to(torch.bfloat16).cuda(); b = torch.randn(N, K).to(torch.bfloat16).cuda()
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda:0")
for _ in range(3):
    ops.gemm(a, b, out, b_t=True)
torch.cuda.synchronize()
r = buf.cpu().view(8, 8)
for w in range(8):
    sL, sW, sB1, sC, sB2, tot, nk = [int(x) for x in r[w][:7]]
    ph = 4 * nk
    print(f"wave {w}: per phase L {sL/ph:7.1f} W {sW/ph:7.1f} B1 {sB1/ph:7.1f} C {sC/ph:7.1f} B2 {sB2/ph:7.1f} | total/phase {tot/ph:7.1f} cycles ({nk} K-tiles)")

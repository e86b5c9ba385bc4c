"""CPU (needs hipcc, no GPU): gemm_bd.hip keeps accumulators and (160-row tiles) the B ring in AGPRs behind the compiler's back, and
loads into VGPRs the compiler does not know to be in flight (256-row tiles).  tools/bd_check_isa.py compiles the file to assembly
and checks, for the four kernel instantiations: no compiler-generated instruction names an AGPR, nothing spills to scratch, and no
compiler-generated instruction touches the destination of an in-flight inline-asm load before the inline-asm wait that covers it."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") and shutil.which("hipcc") is None, reason="hipcc not installed")
def test_b_direct_gemm_isa_invariants():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bd_check_isa.py")], capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if "gemm_bf16_bd_kernel" in ln and "{" in ln]
    assert len(lines) == 4, r.stdout + r.stderr                      # (160, 256 rows) x (row-major, packed B)
    assert all("agpr-outside-asm: 0" in ln and "inflight-vgpr-touched: 0" in ln for ln in lines), r.stdout
    assert all("'ScratchSize': 0" in ln and "'NumAgprs': 256" in ln for ln in lines), r.stdout
    assert r.returncode == 0, r.stdout + r.stderr

"""CPU (needs hipcc, no GPU): gemm_w4.hip keeps its 256 accumulator registers in AGPRs behind the compiler's back.  The
kernels are only correct while no compiler-generated instruction of theirs names an AGPR and nothing spills to scratch;
tools/w4_check_isa.py compiles the file to assembly and checks exactly that for the six kernel instantiations."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") and shutil.which("hipcc") is None, reason="hipcc not installed")
def test_four_wave_gemm_never_lets_the_compiler_touch_agprs():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "w4_check_isa.py")], capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if "gemm_bf16_w4" in l]
    assert len(lines) == 6, r.stdout + r.stderr                      # NT / NN / TN x (LDS-DMA, register-staged)
    lds_dma = [l for l in lines if "w4_kernel" in l]
    assert all("agpr-outside-asm: 0" in l for l in lines), r.stdout
    assert all("'ScratchSize': 0" in l for l in lds_dma), r.stdout   # the register-staged TN build may spill a few bytes
    assert all("'NumAgprs': 256" in l for l in lines), r.stdout

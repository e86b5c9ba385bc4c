"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/afft_hip.h declares.
No compute call is made here (no GPU in the build container)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from afft_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "afft_amd", "csrc"), "-j4"])
    return _lib


def test_header_symbols_are_exported(built_lib):
    hdr = open(os.path.join(ROOT, "include", "afft_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(afft_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 15
    lib = ctypes.CDLL(built_lib.LIB_PATH)
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} declared in include/afft_hip.h but not exported"
    assert sorted(built_lib.EXPORTS) == declared, "ctypes signature table and header disagree"


def test_library_loads_and_reports_version(built_lib):
    assert built_lib.lib().afft_version() >= 1


def test_gemm_desc_matches_c_layout(built_lib):
    # offsets computed by the C compiler for afft_gemm_t must equal the ctypes mirror
    src = r'''
#include <stddef.h>
#include <stdio.h>
#include "afft_hip.h"
int main(){ printf("%zu %zu %zu %zu %zu %zu\n", sizeof(afft_gemm_t), offsetof(afft_gemm_t, alpha),
  offsetof(afft_gemm_t, aux), offsetof(afft_gemm_t, rowscale), offsetof(afft_gemm_t, out), offsetof(afft_gemm_t, out2_dtype)); return 0; }
'''
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(td, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        got = [int(x) for x in subprocess.check_output([exe]).split()]
    G = built_lib.GemmDesc
    want = [ctypes.sizeof(G), G.alpha.offset, G.aux.offset, G.rowscale.offset, G.out.offset, G.out2_dtype.offset]
    assert got == want


@pytest.mark.parametrize("cname,pyname", [("afft_gemm_t", "GemmDesc"), ("afft_dropout_t", "Dropout"), ("afft_sgd_fused_t", "SgdFused"),
                                          ("afft_attn_sublayer_t", "AttnSublayer"), ("afft_mlp_sublayer_t", "MLPSublayer"),
                                          ("afft_cross_attn_sublayer_t", "CrossAttnSublayer")])
def test_every_struct_field_matches_c_layout(built_lib, cname, pyname):
    """sizeof and the offset of EVERY field of each ctypes mirror against what the C compiler lays out for the header."""
    import tempfile
    S = getattr(built_lib, pyname)
    names = [f[0] for f in S._fields_]
    body = "".join(f'printf("%zu\\n", offsetof({cname}, {n}));' for n in names)
    src = f'#include <stddef.h>\n#include <stdio.h>\n#include "afft_hip.h"\nint main(){{ printf("%zu\\n", sizeof({cname})); {body} return 0; }}\n'
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(td, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        got = [int(x) for x in subprocess.check_output([exe]).split()]
    assert got[0] == ctypes.sizeof(S), (got[0], ctypes.sizeof(S))
    for n, off in zip(names, got[1:]):
        assert getattr(S, n).offset == off, (n, getattr(S, n).offset, off)


def test_no_gpu_means_loud_failure(built_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from afft_amd import ops
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.layernorm_fwd(torch.zeros(2, 4), None, None, 1e-6, torch.zeros(2, 4))

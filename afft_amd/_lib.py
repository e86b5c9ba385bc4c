"""ctypes binding of the C-ABI in include/afft_hip.h (libafft_hip.so, built by afft_amd/csrc/Makefile).

The product path fails loudly when the HIP library is missing: there is no CPU or eager-PyTorch
fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AFFT_LIB") or os.path.join(_HERE, "lib", "libafft_hip.so")   # AFFT_LIB: kernel-tuning builds

F32, BF16 = 0, 1
GEMM_WS_HEADER = 4096
ACT_NONE, ACT_GELU_ERF, ACT_GELU_TANH, ACT_DGELU_ERF, ACT_DGELU_TANH, ACT_RELU, ACT_SIGMOID_GATE = 0, 1, 2, 3, 4, 5, 6
MASK_NONE, MASK_DIAG, MASK_CAUSAL, MASK_BLOCKCAUSAL = 0, 1, 2, 3

i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class Dropout(C.Structure):
    _fields_ = [("p", f32), ("key", C.c_uint32), ("path_p", f32), ("path_key", C.c_uint32), ("path_group", i32)]


class GemmDesc(C.Structure):
    _fields_ = [
        ("M", i32), ("N", i32), ("K", i32), ("dtype", i32),
        ("A", vp), ("a_rs", i64), ("a_cs", i64),
        ("B", vp), ("b_rs", i64), ("b_cs", i64),
        ("alpha", f32),
        ("bias", vp),
        ("act", i32),
        ("aux", vp), ("ldaux", i64), ("aux_dtype", i32),
        ("pre", vp), ("ldpre", i64), ("pre_dtype", i32),
        ("rowscale", vp),
        ("residual", vp), ("ldres", i64),
        ("accumulate", i32),
        ("out", vp), ("ldo", i64), ("out_dtype", i32),
        ("out2", vp), ("ldo2", i64), ("out2_dtype", i32),
        ("drop", Dropout),
        ("workspace", vp), ("workspace_bytes", i64),
        ("max_workgroups", i32),
        ("split3", i32), ("a_lo", i64), ("b_lo", i64),
    ]


_SIGS = {
    "afft_version": ([], C.c_int),
    "afft_gemm": ([C.POINTER(GemmDesc), vp], C.c_int),
    "afft_set_gemm_variant": ([C.c_int], C.c_int),
    "afft_set_gemm_splitk": ([C.c_int], C.c_int),
    "afft_gemm_variant_for": ([C.c_int, C.c_int, C.c_int, C.c_int, C.c_int], C.c_int),
    "afft_gemm_splitk_for": ([C.c_int, C.c_int, C.c_int, C.c_int, C.c_int], C.c_int),
    "afft_gemm_workspace_bytes": ([C.c_int, C.c_int, C.c_int, C.c_int, C.c_int], i64),
    "afft_split_bf16": ([vp, i64, i32, i32, vp, i64, i32, i64, vp], C.c_int),
    "afft_layernorm_fwd": ([vp, i64, vp, vp, f32, i32, i32, vp, i64, i32, vp, vp, vp], C.c_int),
    "afft_layernorm_bwd_nparts": ([i32], C.c_int),
    "afft_layernorm_bwd": ([vp, i64, i32, vp, i64, vp, vp, vp, i32, i32, vp, vp, i64, vp, C.POINTER(Dropout), vp, vp, i32,
                            vp, i32, vp, vp], C.c_int),
    "afft_attention_fwd": ([vp, i64, vp, i64, vp, i64, i32, i32, i32, i32, i32, f32, i32, i32, f32, C.c_uint32,
                            vp, i64, vp, vp], C.c_int),
    "afft_attention_bwd": ([vp, i64, vp, i64, vp, i64, vp, i64, i32, vp, i32, i32, i32, i32, f32, f32, C.c_uint32,
                            vp, i64, vp, i64, vp, i64, vp], C.c_int),
    "afft_softmax_ce": ([vp, i64, i32, i32, vp, vp, i64, vp, f32, vp, vp, vp, i64, i32, vp, vp], C.c_int),
    "afft_mse": ([vp, i64, vp, i64, i32, i32, f32, vp, f32, vp, vp, i64, vp, i64, vp], C.c_int),
    "afft_cast": ([vp, i64, i32, i32, vp, i64, i32, vp, i64, i32, C.POINTER(Dropout), vp], C.c_int),
    "afft_assemble_tokens": ([C.POINTER(vp), C.POINTER(i64), i32, vp, i64, vp, i32, i32, i32, vp, vp], C.c_int),
    "afft_colsum": ([vp, i64, i32, i32, i32, vp, i32, vp], C.c_int),
    "afft_add_rows_periodic": ([vp, i64, vp, i64, i32, i32, i32, vp, i64, vp], C.c_int),
    "afft_reduce_rows_periodic": ([vp, i64, i32, i32, i32, vp, i64, vp], C.c_int),
    "afft_sgd_nesterov": ([vp, vp, i32, vp, vp, i64, f32, f32, f32, f32, vp, i32, vp], C.c_int),
    "afft_sumsq": ([vp, i32, i64, f32, vp, vp], C.c_int),
    "afft_group_sum": ([vp, i32, i32, i64, f32, vp, vp], C.c_int),
    "afft_set_dropout_salt": ([vp], C.c_int),
    "afft_dropout_salt_step": ([vp, vp], C.c_int),
    "afft_mixup_plan": ([vp, i32, i32, i64, vp, vp, vp], C.c_int),
    "afft_mixup_rows": ([vp, i32, i64, vp, f32, vp, vp], C.c_int),
    "afft_mixup_labels": ([vp, i32, i32, i32, f32, i64, vp, f32, vp, vp], C.c_int),
    "afft_softmax_rows": ([vp, i64, i32, i32, vp, i64, vp], C.c_int),
    "afft_zero_mask_frames": ([vp, i32, i32, i64, i32, C.c_uint32, vp], C.c_int),
    "afft_act_bwd": ([i32, vp, i64, vp, i64, i32, vp, i64, i32, i32, C.POINTER(Dropout), vp, i64, i32, vp, i64, vp], C.c_int),
    "afft_softmax_small_fwd": ([vp, i64, i32, i32, vp, i64, vp], C.c_int),
    "afft_softmax_small_bwd": ([vp, i64, vp, i64, i32, i32, vp, i64, vp], C.c_int),
    "afft_weighted_sum_fwd": ([C.POINTER(vp), i64, vp, i64, i32, i32, i32, vp, i64, vp], C.c_int),
    "afft_weighted_sum_bwd": ([C.POINTER(vp), i64, vp, i64, vp, i64, i32, i32, i32, C.POINTER(vp), i64, vp, i64, vp], C.c_int),
    "afft_group_bcast": ([vp, i32, i32, i64, f32, vp, vp], C.c_int),
    "afft_clip_coef": ([vp, f32, vp, vp, vp], C.c_int),
}

EXPORTS = sorted(list(_SIGS) + ["afft_last_error"])

_lib = None


def lib():
    """The loaded shared library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"afft_amd: {LIB_PATH} is missing. Build it with `make -C afft_amd/csrc` "
                f"(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
        _lib = C.CDLL(LIB_PATH)
        _lib.afft_last_error.restype = C.c_char_p
        _lib.afft_last_error.argtypes = []
        for name, (args, res) in _SIGS.items():
            fn = getattr(_lib, name)
            fn.argtypes = args
            fn.restype = res
        if os.environ.get("AFFT_GEMM_VARIANT"):
            check(_lib.afft_set_gemm_variant(int(os.environ["AFFT_GEMM_VARIANT"])), "set_gemm_variant")
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().afft_last_error().decode(errors="replace")
        raise RuntimeError(f"afft_hip {what} failed (code {rc}): {msg}")

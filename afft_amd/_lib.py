"""ctypes binding of the C-ABI in include/afft_hip.h (libafft_hip.so, built by afft_amd/csrc/Makefile).

The product path fails loudly when the HIP library is missing: there is no CPU or eager-PyTorch
fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AFFT_LIB") or os.path.join(_HERE, "lib", "libafft_hip.so")   # AFFT_LIB: kernel-tuning builds

F32, BF16, F16 = 0, 1, 2      # F16: fp16 planes / images of the "fp16x2" precision (outputs and copies; never a GEMM operand dtype)
GEMM_WS_HEADER = 4096
ACT_NONE, ACT_GELU_ERF, ACT_GELU_TANH, ACT_DGELU_ERF, ACT_DGELU_TANH, ACT_RELU, ACT_SIGMOID_GATE = 0, 1, 2, 3, 4, 5, 6
MASK_NONE, MASK_DIAG, MASK_CAUSAL, MASK_BLOCKCAUSAL = 0, 1, 2, 3

i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class Dropout(C.Structure):
    _fields_ = [("p", f32), ("key", C.c_uint32), ("path_p", f32), ("path_key", C.c_uint32), ("path_group", i32)]


class SgdFused(C.Structure):      # afft_sgd_fused_t
    _fields_ = [("p", vp), ("buf", vp), ("p_bf16", vp), ("lr", f32), ("mom", f32), ("wd", f32), ("gscale", f32), ("first_step", i32),
                ("p_pk16", vp), ("p_f16", vp), ("p_f8", vp), ("ok", vp)]


SgdP = C.POINTER(SgdFused)


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
        ("split3", i32), ("a_lo", i64), ("b_lo", i64),
        ("sgd", SgdP),
        ("b_packed", vp),
        ("out_lo", i64),
        ("a8", vp), ("a8_ld", i64), ("b8", vp), ("b8_ld", i64),
        ("out_lo8", vp),
    ]


u32, fp = C.c_uint32, C.POINTER(C.c_float)
DropP = C.POINTER(Dropout)
_WS = [("gemm_ws", vp), ("gemm_ws_bytes", i64), ("gemm_ws_aux", vp), ("gemm_ws_aux_bytes", i64)]
_HAND = [("dx_bf16", vp), ("up_drop", DropP), ("up_dcol", vp)]


class AttnSublayer(C.Structure):      # afft_attn_sublayer_t
    _fields_ = [
        ("rows", i32), ("d", i32), ("L", i32), ("H", i32),
        ("conv1d", i32), ("mask", i32), ("mask_period", i32),
        ("eps", f32), ("scale", f32),
        ("x", vp), ("ln_w", vp), ("ln_b", vp),
        ("w_qkv", vp), ("ldw_qkv", i64), ("b_qkv", vp),
        ("w_proj", vp), ("ldw_proj", i64), ("b_proj", vp),
        ("p_attn", f32), ("k_attn", u32),
        ("out_drop", Dropout),
        ("xn", vp), ("qkv", vp), ("ao", vp),
        ("mean", vp), ("rstd", vp), ("probs", vp), ("y", vp),
        ("dy", vp), ("dya", vp), ("dya_ready", i32),
        ("dao", vp), ("dqkv", vp), ("dxn", vp),
        ("g_w_qkv", vp), ("acc_w_qkv", i32), ("g_b_qkv", vp), ("acc_b_qkv", i32),
        ("g_w_proj", vp), ("acc_w_proj", i32), ("g_b_proj", vp), ("acc_b_proj", i32),
        ("g_ln_w", vp), ("g_ln_b", vp), ("acc_ln", i32),
        ("dx", vp)] + _HAND + [("ln_partial", vp)] + _WS + [("sgd_w_qkv", SgdP), ("sgd_w_proj", SgdP), ("w_qkv_pk", vp), ("w_proj_pk", vp),
                                                       ("f16x2", i32), ("xn_b", vp), ("qkv_b", vp), ("ao_b", vp), ("w_qkv8", vp), ("w_proj8", vp), ("take", i32)]


class MLPSublayer(C.Structure):       # afft_mlp_sublayer_t
    _fields_ = [
        ("rows", i32), ("d", i32), ("hidden", i32), ("conv1d", i32),
        ("gelu", i32), ("eps", f32),
        ("x", vp), ("ln_w", vp), ("ln_b", vp),
        ("w1", vp), ("ldw1", i64), ("b1", vp),
        ("w2", vp), ("ldw2", i64), ("b2", vp),
        ("out_drop", Dropout),
        ("xn", vp), ("u", vp), ("h", vp),
        ("mean", vp), ("rstd", vp), ("y", vp),
        ("dy", vp), ("dya", vp), ("dya_ready", i32),
        ("du", vp), ("dxn", vp),
        ("g_w1", vp), ("acc_w1", i32), ("g_b1", vp), ("acc_b1", i32),
        ("g_w2", vp), ("acc_w2", i32), ("g_b2", vp), ("acc_b2", i32),
        ("g_ln_w", vp), ("g_ln_b", vp), ("acc_ln", i32),
        ("dx", vp)] + _HAND + [("ln_partial", vp)] + _WS + [("sgd_w1", SgdP), ("sgd_w2", SgdP), ("w1_pk", vp), ("w2_pk", vp),
                                                       ("f16x2", i32), ("xn_b", vp), ("h_b", vp), ("w1_8", vp), ("w2_8", vp)]


class CrossAttnSublayer(C.Structure):  # afft_cross_attn_sublayer_t
    _fields_ = [
        ("rows", i32), ("d", i32), ("L", i32), ("H", i32), ("mask", i32), ("mask_period", i32),
        ("eps", f32), ("scale", f32),
        ("x", vp), ("mem", vp),
        ("nq_w", vp), ("nq_b", vp), ("nkv_w", vp), ("nkv_b", vp),
        ("w_q", vp), ("w_k", vp), ("w_v", vp), ("w_proj", vp), ("ldw", i64),
        ("b_proj", vp),
        ("p_attn", f32), ("k_attn", u32), ("out_drop", Dropout),
        ("xq", vp), ("mkv", vp), ("q", vp), ("k", vp), ("v", vp), ("ao", vp),
        ("mean_q", vp), ("rstd_q", vp), ("mean_kv", vp), ("rstd_kv", vp), ("probs", vp),
        ("y", vp),
        ("dy", vp), ("dya", vp), ("dya_ready", i32),
        ("dao", vp), ("dq", vp), ("dk", vp), ("dv", vp), ("dxq", vp),
        ("dmkv", vp),
        ("g_w_q", vp), ("acc_w_q", i32), ("g_w_k", vp), ("acc_w_k", i32), ("g_w_v", vp), ("acc_w_v", i32),
        ("g_w_proj", vp), ("acc_w_proj", i32), ("g_b_proj", vp), ("acc_b_proj", i32),
        ("g_nq_w", vp), ("g_nq_b", vp), ("acc_nq", i32), ("g_nkv_w", vp), ("g_nkv_b", vp), ("acc_nkv", i32),
        ("dx", vp), ("dmem", vp)] + _HAND + [("ln_partial", vp), ("ln_partial2", vp)] + _WS + [
            ("sgd_w_q", SgdP), ("sgd_w_k", SgdP), ("sgd_w_v", SgdP), ("sgd_w_proj", SgdP)]


class GemmTraceRec(C.Structure):   # afft_gemm_trace_rec_t
    _fields_ = [("M", i32), ("N", i32), ("K", i32), ("a_kstrided", i32), ("b_kstrided", i32), ("variant", i32), ("splitk", i32),
                ("split3", i32), ("fused_update", i32), ("ms", f32)]


class KernelTraceRec(C.Structure):   # afft_kernel_trace_rec_t
    _fields_ = [("kind", i32), ("rows", i32), ("width", i32), ("reserved", i32), ("bytes", i64), ("flops", i64), ("ms", f32)]


K_ATTN_FWD, K_ATTN_BWD, K_LN_FWD, K_LN_BWD = 1, 2, 3, 4

_SIGS = {
    "afft_version": ([], C.c_int),
    "afft_gemm": ([C.POINTER(GemmDesc), vp], C.c_int),
    "afft_set_gemm_variant": ([C.c_int], C.c_int),
    "afft_set_gemm_splitk": ([C.c_int], C.c_int),
    "afft_gemm_variant_for": ([C.c_int, C.c_int, C.c_int, C.c_int, C.c_int], C.c_int),
    "afft_gemm_splitk_for": ([C.c_int, C.c_int, C.c_int, C.c_int, C.c_int], C.c_int),
    "afft_gemm_trace_begin": ([i32], C.c_int),
    "afft_gemm_trace_end": ([C.POINTER(GemmTraceRec), i32], C.c_int),
    "afft_kernel_trace_begin": ([i32], C.c_int),
    "afft_kernel_trace_end": ([C.POINTER(KernelTraceRec), i32], C.c_int),
    "afft_gemm_workspace_bytes": ([C.c_int, C.c_int, C.c_int, C.c_int, C.c_int], i64),
    "afft_gemm_packed_wanted": ([C.c_int, C.c_int, C.c_int], C.c_int),
    "afft_pack_weight": ([vp, i64, i32, i32, vp, vp], C.c_int),
    "afft_split_bf16": ([vp, i64, i32, i32, vp, i64, i32, i64, vp], C.c_int),
    "afft_split_f16": ([vp, i64, i32, i32, vp, i64, i32, i64, vp], C.c_int),
    "afft_layernorm_fwd": ([vp, i64, vp, vp, f32, i32, i32, vp, i64, i32, vp, vp, vp], C.c_int),
    "afft_layernorm_fwd_split": ([vp, i64, vp, vp, f32, i32, i32, vp, i64, i64, vp, i64, vp, vp, vp, vp], C.c_int),
    "afft_quant_e4m3": ([vp, i64, i32, i32, f32, vp, i64, i32, vp, vp], C.c_int),
    "afft_gemm_lo8_ok": ([C.c_int, C.c_int, C.c_int], C.c_int),
    "afft_layernorm_bwd_nparts": ([i32], C.c_int),
    "afft_layernorm_bwd": ([vp, i64, i32, vp, i64, vp, vp, vp, i32, i32, vp, vp, i64, vp, C.POINTER(Dropout), vp, vp, i32,
                            vp, i32, vp, vp], C.c_int),
    "afft_attention_fwd": ([vp, i64, vp, i64, vp, i64, i32, i32, i32, i32, i32, f32, i32, i32, f32, C.c_uint32,
                            vp, i64, vp, vp], C.c_int),
    "afft_attention_fwd_table": ([vp, i64, vp, i64, vp, i64, i32, i32, i32, i32, i32, f32, vp, f32, C.c_uint32, vp, i64, vp, vp], C.c_int),
    "afft_attention_fwd_split": ([vp, i64, vp, i64, vp, i64, i64, i32, i32, i32, i32, f32, i32, i32, f32, C.c_uint32,
                                  vp, i64, i64, vp, i64, vp, vp, vp], C.c_int),
    "afft_attention_bwd": ([vp, i64, vp, i64, vp, i64, vp, i64, i32, vp, i32, i32, i32, i32, f32, f32, C.c_uint32,
                            vp, i64, vp, i64, vp, i64, vp], C.c_int),
    "afft_softmax_ce": ([vp, i64, i32, i32, vp, vp, i64, vp, f32, vp, vp, vp, i64, i32, vp, vp], C.c_int),
    "afft_loss_reduce": ([C.POINTER(vp), C.POINTER(i64), C.POINTER(f32), i32, vp, vp, vp], C.c_int),
    "afft_loss_reduce_bwd_ok": ([C.POINTER(vp), C.POINTER(i64), C.POINTER(f32), i32, vp, vp, vp, vp], C.c_int),
    "afft_loss_reduce_bwd": ([C.POINTER(vp), C.POINTER(i64), C.POINTER(f32), i32, vp, vp], C.c_int),
    "afft_layernorm_bwd_take": ([vp, i64, i32, vp, i64, vp, vp, vp, i32, i32, vp, i64, i32, vp, i64, vp, vp, vp, vp, i32, vp, i32, vp, vp], C.c_int),
    "afft_softmax_ce_frames": ([vp, i64, i64, i32, i32, i32, vp, vp, i64, vp, f32, vp, vp, i64, i64, i32, vp, vp], C.c_int),
    "afft_mse_loss": ([vp, i64, vp, i64, i32, i32, f32, vp, vp, i64, vp], C.c_int),
    "afft_mse_frames_bwd": ([vp, i64, i32, i32, vp, i64, i32, i32, i32, i32, f32, vp, vp, vp, vp], C.c_int),
    "afft_mse": ([vp, i64, vp, i64, i32, i32, f32, vp, f32, vp, vp, i64, vp, i64, vp, i64, vp], C.c_int),
    "afft_cast": ([vp, i64, i32, i32, vp, i64, i32, vp, i64, i32, C.POINTER(Dropout), vp], C.c_int),
    "afft_assemble_tokens": ([C.POINTER(vp), C.POINTER(i64), i32, vp, i64, vp, i32, i32, i32, vp, vp], C.c_int),
    "afft_colsum": ([vp, i64, i32, i32, i32, vp, i32, vp, i64, vp], C.c_int),
    "afft_zero": ([vp, i64, vp], C.c_int),
    "afft_gather_frames": ([vp, i64, i64, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp], C.c_int),
    "afft_add_rows_periodic": ([vp, i64, vp, i64, i32, i32, i32, vp, i64, vp], C.c_int),
    "afft_reduce_rows_periodic": ([vp, i64, i32, i32, i32, vp, i64, vp], C.c_int),
    "afft_sgd_nesterov": ([vp, vp, i32, vp, vp, i64, f32, f32, f32, f32, vp, i32, vp], C.c_int),
    "afft_sgd_nesterov_runs": ([vp, vp, vp, vp, vp, i32, f32, f32, f32, f32, i32, vp], C.c_int),
    "afft_sgd_nesterov2": ([vp, vp, i32, vp, vp, vp, vp, i64, f32, f32, f32, f32, vp, i32, vp, vp], C.c_int),
    "afft_sgd_nesterov_runs2": ([vp, vp, vp, vp, vp, vp, vp, i32, f32, f32, f32, f32, i32, vp, vp], C.c_int),
    "afft_sumsq": ([vp, i32, i64, f32, vp, vp, i64, vp], C.c_int),
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
    "afft_attn_sublayer_fwd": ([C.POINTER(AttnSublayer), vp], C.c_int),
    "afft_attn_sublayer_bwd": ([C.POINTER(AttnSublayer), vp, vp], C.c_int),
    "afft_mlp_sublayer_fwd": ([C.POINTER(MLPSublayer), vp], C.c_int),
    "afft_mlp_sublayer_bwd": ([C.POINTER(MLPSublayer), vp, vp], C.c_int),
    "afft_cross_attn_sublayer_fwd": ([C.POINTER(CrossAttnSublayer), vp], C.c_int),
    "afft_cross_attn_sublayer_bwd": ([C.POINTER(CrossAttnSublayer), vp, vp], C.c_int),
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
        if os.environ.get("AFFT_GEMM_SPLITK"):      # 0 off, 1 automatic (default), 2 / 4 forced
            check(_lib.afft_set_gemm_splitk(int(os.environ["AFFT_GEMM_SPLITK"])), "set_gemm_splitk")
        if os.environ.get("AFFT_GEMM_VARIANT"):
            check(_lib.afft_set_gemm_variant(int(os.environ["AFFT_GEMM_VARIANT"])), "set_gemm_variant")
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().afft_last_error().decode(errors="replace")
        raise RuntimeError(f"afft_hip {what} failed (code {rc}): {msg}")

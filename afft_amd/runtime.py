"""Host-side runtime shared by the mirrored reference modules: precision mode, cached bf16 weight
images for the MFMA fast path, and the gradient sink that lets weight-gradient GEMMs accumulate
straight into (flat, all-reduce-friendly) ``.grad`` storage.

Precision modes
  * ``bf16`` (default, the speed mode BASELINE.json's configs[1] names): bf16 operands on
    v_mfma_f32_16x16x32_bf16, fp32 accumulation, fp32 residual stream / LayerNorm statistics /
    softmax / losses / master weights / gradients.
  * ``fp32`` (parity mode): every GEMM on the exact-fp32 MFMA path, all activations fp32; this is
    the mode whose logits are checked to 1e-3 against the reference's fp32 CPU path.
  * ``bf16x3`` (fast parity mode, SURVEY.md 7 hard part 1): activations, residual stream and every non-GEMM kernel
    as in ``fp32``; each GEMM splits its fp32 operands into two bf16 planes (x = hi + lo) and accumulates
    hi*hi + lo*hi + hi*lo in ONE launch of the bf16 MFMA kernels (afft_gemm_t.split3): products exact to
    ~2^-17, logits within the 1e-3 tolerance at roughly a third of the bf16 mode's GEMM rate instead of 1/16.
"""
from __future__ import annotations

import os
import weakref
from typing import Dict, Optional

import torch

from . import ops

Tensor = torch.Tensor

_PRECISION = os.environ.get("AFFT_PRECISION", "bf16")
_GRAD_MODE = os.environ.get("AFFT_GRAD_MODE", "sink")  # 'sink' | 'autograd'


PRECISIONS = ("bf16", "fp32", "bf16x3", "fp16x2")


def set_precision(p: str):
    global _PRECISION
    if p not in PRECISIONS:
        raise ValueError("precision must be one of " + ", ".join(repr(x) for x in PRECISIONS))
    _PRECISION = p


def precision() -> str:
    return _PRECISION


class precision_scope:
    """`with precision_scope("fp16x2"):` -- run a block in another precision and come back (None: leave it as it is)"""

    def __init__(self, p: Optional[str]):
        self.p, self.saved = p, None

    def __enter__(self):
        if self.p is not None:
            self.saved = _PRECISION
            set_precision(self.p)
        return self

    def __exit__(self, *exc):
        if self.saved is not None:
            set_precision(self.saved)
        return False


def backward_precision() -> Optional[str]:
    """The precision a backward pass runs in when it differs from the forward's: 'fp16x2' is a FORWARD format (activations hi + lo
    in fp16, weights rounded once to fp16: logits inside the reference's 1e-3); its backward pass is the single-pass bf16 one on
    bf16 copies of the saved activations and the bf16 weight images (gradients keep fp32's exponent range: no loss scaling)."""
    return "bf16" if _PRECISION == "fp16x2" else None


def act_dtype() -> torch.dtype:
    return torch.bfloat16 if _PRECISION == "bf16" else torch.float32


def fp32_acts() -> bool:
    """activations are kept in fp32 (the parity modes 'fp32', 'bf16x3' and 'fp16x2')"""
    return _PRECISION != "bf16"


def split_mode() -> Optional[str]:
    """how the GEMM operands of the current precision are split into 16-bit planes: None, 'bf16' (bf16x3) or 'f16' (fp16x2)"""
    return {"bf16x3": "bf16", "fp16x2": "f16"}.get(_PRECISION)


def set_grad_mode(m: str):
    """'sink': weight gradients are accumulated by the wgrad GEMM epilogue directly into param.grad
    (no autograd accumulation, no per-parameter hooks; use afft_amd.parallel.GradReducer for DP).
    'autograd': weight gradients are returned through autograd (works under torch DDP, slower)."""
    global _GRAD_MODE
    if m not in ("sink", "autograd"):
        raise ValueError(m)
    _GRAD_MODE = m


def grad_mode() -> str:
    return _GRAD_MODE


def pad64(n: int) -> int:
    return (n + 63) // 64 * 64


# --------------------------------------------------------------------------- bf16 weight images
class _WImage:
    __slots__ = ("version", "ptr", "w", "external", "pk", "pk_live", "pk_version", "h", "h_version", "h_external", "e", "e_version",
                 "e_external")


_wlist: list = []   # weak references to parameters that own an image (for invalidate_weight_images)


def _register(p: Tensor):
    """ONE weak reference per parameter, however often its image / split is rebuilt (a parameter whose split is dropped
    after every optimizer step must not grow the list by an entry per step)"""
    if not getattr(p, "_afft_listed", False):
        _wlist.append(weakref.ref(p))
        p._afft_listed = True


def weight_images(p: Tensor) -> Tensor:
    """w16: the bf16 image [pad64(rows), pad64(cols)] of a 2-D fp32 parameter, zero padded (the k-contiguous "NT" operand of
    the forward GEMM of nn.Linear / the data-gradient GEMM of HF Conv1D, the k-strided "NN" operand of the other two).
    Refreshed by one cast kernel when the parameter's version counter or storage changes; parameters re-homed by
    afft_amd.parallel.FlatParams have it written by the optimizer kernels (`external`)."""
    img = getattr(p, "_afft_img", None)
    ver = p._version
    if img is None or img.ptr != p.data_ptr() or img.w.device != p.device:
        img = _WImage()
        rows, cols = p.shape
        img.w = torch.zeros(pad64(rows), pad64(cols), dtype=torch.bfloat16, device=p.device)
        img.version = -1
        img.external = False
        img.pk = None
        img.pk_live = False
        img.pk_version = -1
        img.h = None
        img.h_version = -1
        img.h_external = False
        img.e = None
        img.e_version = -1
        img.e_external = False
        img.ptr = p.data_ptr()
        p._afft_img = img
        _register(p)
    if img.version != ver:
        with torch.no_grad():
            ops.cast(p.detach(), img.w[:p.shape[0], :p.shape[1]])
        img.version = ver
    return img.w


def weight_f16(p: Tensor) -> Tensor:
    """FP16 image [pad64(rows), pad64(cols)] of a 2-D fp32 parameter: the B operand of the fp16 two-pass forward GEMMs ('fp16x2'
    precision; the weight is rounded ONCE to fp16, the activation side carries hi + lo).  A parameter homed in the flat buffers
    (parallel.FlatParams) has its image there, written by the optimizer kernels beside the bf16 one (FlatParams.f16_images);
    any other parameter gets a cast image that is redone when its version counter or storage changes."""
    weight_images(p)                      # creates / validates the record (and the bf16 image the backward pass reads)
    img = p._afft_img
    if img.h_external:
        if img.h_version != p._version:   # written from outside (load_state_dict, p.copy_): re-derive, as weight_images does
            with torch.no_grad():
                ops.cast(p.detach(), img.h)
            img.h_version = p._version
        return img.h
    if img.h is None:
        rows, cols = p.shape
        img.h = torch.zeros(pad64(rows), pad64(cols), dtype=torch.float16, device=p.device)
        img.h_version = -1
    if img.h_version != p._version:
        with torch.no_grad():
            ops.cast(p.detach(), img.h[:p.shape[0], :p.shape[1]])
        img.h_version = p._version
    return img.h


_LO8 = True


def lo8() -> bool:
    """'fp16x2' forward: run the second pass A_lo W of the big nn.Linear GEMMs on the block-scaled fp8 MFMA (afft_gemm_t.split3 = 3:
    A_lo and W as e4m3 bytes with constant block scales, twice the bf16 rate) instead of a second fp16 pass.  The term it computes is
    ~2^-12 of the product, so its 2^-4 operand rounding is ~2^-16 of the result -- below the fp16 rounding of the weight the mode
    already carries.  set_lo8(False): both passes in fp16."""
    return _LO8


def set_lo8(on: bool):
    global _LO8
    _LO8 = bool(on)


ONE_PASS_SITES = ("linear.qkv", "linear.attn", "linear.proj", "linear.fc1", "linear.fc2", "conv1d.qkv", "conv1d.attn", "conv1d.proj", "conv1d.fc1", "conv1d.fc2")


def _parse_sites(text: str) -> frozenset:
    names = [t.strip() for t in text.split(",") if t.strip()]
    out = set()
    for n in names:
        hit = [s for s in ONE_PASS_SITES if s == n or s.startswith(n + ".") or s.endswith("." + n)]
        if not hit:
            raise ValueError(f"afft_amd: unknown one-pass site {n!r}; sites: {', '.join(ONE_PASS_SITES)} (or 'linear', 'conv1d', 'qkv', ...)")
        out.update(hit)
    return frozenset(out)


# default: the predictor's sub-layers (M = B*T rows: their second pass is a full fp16 pass on the 128x128 kernel, +55 % per GEMM, and their
# operand rounding barely reaches the logits) and the fusers' fc2 (K = 4d: the most expensive lo pass and lo plane).  cfg2: logits 6.6e-4 ->
# 7.5e-4 of the 1e-3 tolerance, step 17.70 -> 16.54 ms on one box; EK100 widths 7.1e-4 -> 8.1e-4 (profiles/r06_lo_pass_sweep.txt).
# AFFT_ONE_PASS_SITES= (empty): every site two passes.
_ONE_PASS = _parse_sites(os.environ.get("AFFT_ONE_PASS_SITES", "conv1d.qkv,conv1d.proj,conv1d.fc1,conv1d.fc2,linear.fc2"))


def one_pass_sites() -> frozenset:
    """'fp16x2' forward: the GEMM sites of the composite sub-layers that read their activation as ONE fp16 plane (afft_gemm_t.split3 = 4,
    AFFT_F16X2_ONE_PASS_*) instead of hi + lo -- '<linear|conv1d>.<qkv|attn|proj|fc1|fc2>': nn.Linear sub-layers (the fusers) or GPT-2 Conv1D
    ones (the predictor); 'attn' = the attention core's q, k, v.  Each such site adds one fp16 operand rounding (what its weight already carries) to the forward; how many the 1e-3
    logits tolerance affords is measured by tools/lo_pass_sweep.py."""
    return _ONE_PASS


def set_one_pass_sites(sites):
    global _ONE_PASS
    _ONE_PASS = _parse_sites(sites) if isinstance(sites, str) else _parse_sites(",".join(sites))


_ONE_PASS_MIN_DIM = int(os.environ.get("AFFT_ONE_PASS_MIN_DIM", "1024"))


def set_one_pass_min_dim(width: int) -> int:
    """the model width from which one_pass_sites() apply (tests run the sites on the small goldens with 0); returns the previous value"""
    global _ONE_PASS_MIN_DIM
    prev, _ONE_PASS_MIN_DIM = _ONE_PASS_MIN_DIM, int(width)
    return prev


def one_pass_flags(conv1d: bool, width: int, first: str, second: str, core: str = "") -> int:
    """the AFFT_F16X2_ONE_PASS_* flags of a composite sub-layer of model width `width`: its sites in one_pass_sites(), taken only where the
    second pass costs time -- width >= AFFT_ONE_PASS_MIN_DIM (1024); below (the small test models, whose fp16 weight rounding alone sits at
    ~9e-4 of the 1e-3 tolerance) a lo pass is microseconds and every site keeps it"""
    if width < _ONE_PASS_MIN_DIM:
        return 0
    kind = "conv1d." if conv1d else "linear."
    return ((4 if kind + first in _ONE_PASS else 0) | (8 if kind + second in _ONE_PASS else 0)      # AFFT_F16X2_ONE_PASS_1 | _2 | _ATTN
            | (16 if core and kind + core in _ONE_PASS else 0))


def weight_f8(p: Tensor) -> Tensor:
    """e4m3 byte image e4m3(2^8 p) [pad64(rows), pad64(cols)] of a 2-D parameter (afft_gemm_t.b8): in the flat buffers when the parameter
    is homed there (written by the optimizer kernels), else a quantised copy redone when the parameter changes"""
    weight_images(p)
    img = p._afft_img
    if img.e_external:
        if img.e_version != p._version:
            with torch.no_grad():
                ops.quant_e4m3(p.detach(), 256.0, img.e)
            img.e_version = p._version
        return img.e
    if img.e is None:
        rows, cols = p.shape
        img.e = torch.zeros(pad64(rows), pad64(cols), dtype=torch.uint8, device=p.device)
        img.e_version = -1
    if img.e_version != p._version:
        with torch.no_grad():
            ops.quant_e4m3(p.detach(), 256.0, img.e)
        img.e_version = p._version
    return img.e


def adopt_weight_f8(p: Tensor, view8: Tensor):
    img = p._afft_img
    img.e, img.e_version, img.e_external = view8, p._version, True


def adopt_weight_f16(p: Tensor, view16: Tensor):
    """use `view16` (fp16, the shape of p, kept fresh by the optimizer kernels) as p's FP16 image"""
    img = p._afft_img
    img.h, img.h_version, img.h_external = view16, p._version, True


def weight_split(p: Tensor):
    """bf16x3 mode: the two-plane bf16 split (ops.Split) of a 2-D fp32 parameter, cached until the parameter changes
    (version counter / storage) or invalidate_weight_images() is called (the fused SGD kernel writes through raw
    pointers, so the Trainer invalidates after every step)."""
    ent = getattr(p, "_afft_split", None)
    f16 = split_mode() == "f16"
    if ent is None or ent[0] != p._version or ent[1] != p.data_ptr() or ent[2].f16 != f16:
        with torch.no_grad():
            sp = ops.Split(p.detach(), f16=f16)
        _register(p)
        ent = (p._version, p.data_ptr(), sp)
        p._afft_split = ent
    return ent[2]


_PACKED_IMAGES = os.environ.get("AFFT_PACKED_IMAGES", "1") != "0"


def packed_images() -> bool:
    """Reserve a fragment-packed bf16 slot beside every GEMM weight a Trainer / afft_amd.optim.SGD owns (parallel.FlatParams): the
    forward GEMMs of nn.Linear layers may then run on the "B direct" kernels (csrc/gemm_bd.hip).  +2 bytes per parameter of
    memory; optimizer traffic (+2 B per parameter and step, written by the same epilogues / one pack kernel per weight) only for
    the images that went live (weight_packed)."""
    return _PACKED_IMAGES


def weight_packed(p: Tensor, rows: Optional[int] = None) -> Optional[Tensor]:
    """The fragment-packed image of nn.Linear weight p [out, in] for a forward GEMM over `rows` rows, or None.
    Images come to life on demand: the slot exists for every weight the flat buffers own (parallel.FlatParams), but it is
    filled -- and from then on kept fresh by the optimizer paths, at 2 bytes per parameter and step -- only once the GEMM
    dispatcher says it would use it for a problem of this size (afft_gemm_packed_wanted: at cfg2 the fuser's projection and fc2
    weights, a fifth of the parameters).  rows = None: only report a live image."""
    img = getattr(p, "_afft_img", None)
    if img is None or not img.external or img.pk is None:
        return None
    if not img.pk_live:
        if rows is None or p.dim() != 2:
            return None
        from . import _lib as L_, ops
        if not L_.lib().afft_gemm_packed_wanted(int(rows), int(p.shape[0]), int(p.shape[1])):
            return None
        with torch.no_grad():
            ops.pack_weight(p.detach(), img.pk)      # forward pass, current stream: the parameter is at rest
        img.pk_live = True
        img.pk_version = p._version
    elif img.pk_version != p._version:
        # the parameter was written from outside the optimizer kernels (model.load_state_dict, p.copy_ into the flat views: they
        # bump the version counter; the optimizer kernels write through raw pointers and keep every image fresh themselves):
        # re-pack, exactly as weight_images() re-casts the row-major image
        from . import ops
        with torch.no_grad():
            ops.pack_weight(p.detach(), img.pk)
        img.pk_version = p._version
    return img.pk


def packed_live(p: Tensor) -> bool:
    img = getattr(p, "_afft_img", None)
    return bool(img is not None and img.external and img.pk is not None and img.pk_live)


def adopt_weight_image(p: Tensor, view16: Tensor, packed: Optional[Tensor] = None):
    """Use `view16` (bf16, same shape as p, both dims multiples of 64, kept fresh by the optimizer kernel) as p's MFMA image;
    `packed`: the fragment-packed copy of view16 (ops.pack_weight, afft_gemm_t.b_packed), kept fresh by the same optimizer paths
    (parallel.FlatParams.refresh_packed)."""
    img = _WImage()
    img.w = view16
    img.pk = packed
    img.pk_live = False
    img.pk_version = -1
    img.h = None
    img.h_version = -1
    img.h_external = False
    img.e = None
    img.e_version = -1
    img.e_external = False
    img.version = p._version
    img.external = True
    img.ptr = p.data_ptr()
    p._afft_img = img
    _register(p)


def invalidate_weight_images(include_external: bool = False):
    alive = []
    for r in _wlist:
        p = r()
        if p is None:
            continue
        img = getattr(p, "_afft_img", None)
        if img is not None and (include_external or not img.external):
            img.version = -1
        if img is not None and (include_external or not img.h_external):
            img.h_version = -1
        if img is not None and (include_external or not img.e_external):
            img.e_version = -1
        if img is not None and include_external:
            img.pk_version = -1
        if getattr(p, "_afft_split", None) is not None:
            p._afft_split = None
        if img is not None or hasattr(p, "_afft_split"):
            alive.append(r)
        else:
            p._afft_listed = False
    _wlist[:] = alive


# --------------------------------------------------------------------------- gradient sink
class GradSink:
    """Tracks which parameter gradients have been written during the current backward pass so the first
    wgrad of a step overwrites (no zero-fill pass over ~2.5 GB) and later ones accumulate."""

    def __init__(self):
        self.touched: Dict[int, bool] = {}
        self.on_grad_ready = None  # callback(param) fired after a parameter's gradient has been produced
        self.touch_count: Dict[int, int] = {}   # gradient contributions per parameter in the current step
        self.composite_weights: set = set()     # ids of the weights whose gradient GEMM ran inside a composite backward
        self.fused = None          # callable(param) -> _lib.SgdFused or None: the optimizer fused into that weight's gradient GEMM
        self.fused_applied: Dict[int, int] = {}  # weights whose update ran in a GEMM epilogue in the current step (-> count)
        self.step_ok = None        # device float of the optimizer that owns the running step: the loss reduction writes isfinite(loss)
                                   # there and the update kernels skip a non-finite step (reference: 'The loss is NaN!' before backward)

    def begin_step(self):
        self.touched.clear()
        self.touch_count.clear()
        self.composite_weights.clear()     # ids are only meaningful within the step that recorded them
        self.fused_applied.clear()

    def fused_desc(self, p: Tensor):
        """The fused-update descriptor (afft_sgd_fused_t) for weight p, or None: set by afft_amd.parallel.Trainer for the
        duration of a step whose optimizer runs inside the backward pass (single GPU, no clipping), for the weights it has
        seen receive exactly one gradient contribution per step."""
        self.composite_weights.add(id(p))
        f = self.fused
        return f(p) if f is not None else None

    def grad_buffer(self, p: Tensor):
        """(grad tensor, accumulate?)"""
        if self.fused_applied.get(id(p), 0) > 0:
            raise RuntimeError("fused optimizer: a weight that was already UPDATED in its gradient GEMM's epilogue in this step is "
                               "receiving another gradient contribution (the graph changed since the fused set was learned).  Its "
                               "parameter, momentum and bf16 image have been modified from a partial gradient: reload the last "
                               "checkpoint, then rebuild the Trainer / optimizer or call runtime.set_fused_sgd(False)")
        self.touch_count[id(p)] = self.touch_count.get(id(p), 0) + 1
        if p.grad is None:
            p.grad = torch.zeros_like(p)
            self.touched[id(p)] = True
            return p.grad, True
        first = not self.touched.get(id(p), False)
        self.touched[id(p)] = True
        return p.grad, not first

    def finish_step(self, params):
        """Zero the gradients of parameters that took no part in this step (sink mode only)."""
        if _GRAD_MODE != "sink":
            return
        for p in params:
            if p.requires_grad and p.grad is not None and not self.touched.get(id(p), False):
                p.grad.zero_()


SINK = GradSink()


_AUX_STREAMS: Dict[int, "torch.cuda.Stream"] = {}
_OVERLAP_WGRAD = os.environ.get("AFFT_OVERLAP_WGRAD", "1") != "0"


def new_stream(device, *others) -> "torch.cuda.Stream":
    """A stream from torch's pool (32 per device, handed out round-robin: the 33rd request returns the first stream again) that is
    none of `others` (streams or None) and not this device's auxiliary stream.  Two roles of one step on one HIP stream are legal
    for eager work but not inside a capture: a stream that waits for its own alias made hip::Stream::EndCapture recurse until the
    stack ran out (ROCm 7.0; found in round 5 once the test session had drawn enough streams for the capture stream to come back
    as the auxiliary one)."""
    import torch
    idx = device.index if getattr(device, "index", None) is not None else torch.cuda.current_device()
    taken = {s.cuda_stream for s in others if s is not None}
    if idx in _AUX_STREAMS:
        taken.add(_AUX_STREAMS[idx].cuda_stream)
    taken.add(torch.cuda.current_stream(idx).cuda_stream)
    for _ in range(64):
        st = torch.cuda.Stream(device=idx)
        if st.cuda_stream not in taken:
            return st
    raise RuntimeError("afft_amd.runtime.new_stream: torch's stream pool has no stream left that is distinct from %d others" % len(taken))


_AUX_PRIORITY = os.environ.get("AFFT_AUX_PRIORITY", "normal")      # "low" | "normal" | "high": HIP priority of the auxiliary stream


def priority_stream(idx: int, which: str) -> "torch.cuda.Stream":
    """A HIP stream of the lowest ('low') or highest ('high') priority the device offers, as a torch stream (torch's own pool only
    has normal- and high-priority streams).  Created once per call with hipStreamCreateWithPriority and never destroyed."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    least, greatest = ctypes.c_int(0), ctypes.c_int(0)
    if hip.hipDeviceGetStreamPriorityRange(ctypes.byref(least), ctypes.byref(greatest)) != 0:
        raise RuntimeError("hipDeviceGetStreamPriorityRange failed")
    h = ctypes.c_void_p()
    with torch.cuda.device(idx):
        if hip.hipStreamCreateWithPriority(ctypes.byref(h), 1, least.value if which == "low" else greatest.value) != 0:      # 1 = hipStreamNonBlocking
            raise RuntimeError("hipStreamCreateWithPriority failed")
    return torch.cuda.ExternalStream(h.value, device=idx)


def aux_stream(device) -> "torch.cuda.Stream":
    """Per-device side stream on which weight-gradient GEMMs run beside the data-gradient chain."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _AUX_STREAMS.get(idx)
    if st is None:
        st = priority_stream(idx, _AUX_PRIORITY) if _AUX_PRIORITY in ("low", "high") else new_stream(torch.device("cuda", idx))
        _AUX_STREAMS[idx] = st
    return st


def overlap_wgrad() -> bool:
    return _OVERLAP_WGRAD


def set_overlap_wgrad(on: bool):
    global _OVERLAP_WGRAD
    _OVERLAP_WGRAD = bool(on)


_SKIP_DEAD_ROWS = os.environ.get("AFFT_SKIP_DEAD_ROWS", "1") != "0"


def skip_dead_rows() -> bool:
    """The SA-Fuser returns token 0 of its last block only (models/fusion.py:362-365): with this on (default) that block's
    MLP half runs on the token-0 rows alone instead of all M + 1 tokens of a frame -- rows the reference computes and throws
    away.  Outputs and gradients are unchanged with dropout off (eval, p = 0); with the MLP's element dropout on, the compact rows
    draw their masks at other element indices than the same rows of the full-row run (transformerblock.Block.forward_rows_first_token):
    the training trajectory is equivalent in distribution, not bitwise.  AFFT_SKIP_DEAD_ROWS=0 runs the reference's full row set
    (bench.py --full-rows)."""
    return _SKIP_DEAD_ROWS


def set_skip_dead_rows(on: bool):
    global _SKIP_DEAD_ROWS
    _SKIP_DEAD_ROWS = bool(on)



_COMPOSITE = os.environ.get("AFFT_COMPOSITE", "1") != "0"


def composite() -> bool:
    """Enqueue a whole sub-layer (forward or backward) through ONE composite C-ABI call (afft_*_sublayer_fwd / _bwd,
    include/afft_hip.h) instead of one call per kernel: the same kernels in the same order, a fifth of the host work."""
    return _COMPOSITE


def set_composite(on: bool):
    global _COMPOSITE
    _COMPOSITE = bool(on)


_FUSED_SGD = os.environ.get("AFFT_FUSED_SGD", "1") != "0"


def fused_sgd() -> bool:
    """Let the Trainer fuse the optimizer into the weight-gradient GEMM epilogues (single GPU, no gradient clipping)."""
    return _FUSED_SGD


def set_fused_sgd(on: bool):
    global _FUSED_SGD
    _FUSED_SGD = bool(on)


_FUSE_MIN_ELEMS = int(os.environ.get("AFFT_FUSE_MIN_ELEMS", "0"))


def fuse_min_elems() -> int:
    """Smallest weight (elements) whose update runs inside its own weight-gradient GEMM epilogue; smaller weights are left to the per-bucket
    update kernel.  The epilogue's optimizer traffic (18 B per parameter) is carried by the GEMM's own workgroups: a weight with fewer
    tiles than the chip has CUs streams it from a fraction of the chip, while the bucket kernel spreads the same bytes over all of it."""
    return _FUSE_MIN_ELEMS


def set_fuse_min_elems(n: int):
    global _FUSE_MIN_ELEMS
    _FUSE_MIN_ELEMS = int(n)


CAPTURING = False          # a hipGraph capture of the step is under way (afft_amd.parallel.Trainer.capture)
KEEPALIVE: list = []       # tensors read on the auxiliary stream during a capture: kept until the capture ends, because
                           # inside a capture the allocator would hand their memory to a later main-stream allocation


_HANDOVER = os.environ.get("AFFT_HANDOVER", "1") != "0"


def handover() -> bool:
    """Let a sub-layer's LayerNorm-backward kernel also emit the bf16 operand and the output-bias gradient of the
    sub-layer upstream of it (afft_amd/functional.py, "gradient hand-over"); off: separate cast / column-sum kernels."""
    return _HANDOVER


def set_handover(on: bool):
    global _HANDOVER
    _HANDOVER = bool(on)


def empty(*shape, dtype=torch.float32, device=None) -> Tensor:
    return torch.empty(*shape, dtype=dtype, device=device)


def padded_rows(rows: int, cols: int, dtype, device, zero_tail: bool = True) -> Tensor:
    """[pad64(rows), cols] buffer whose tail rows are zero: lets a wgrad (TN) GEMM reduce over the row
    dimension in whole 64-deep steps.  Returns the full padded buffer; use [:rows] for the live part."""
    pr = pad64(rows)
    t = torch.empty(pr, cols, dtype=dtype, device=device)
    if zero_tail and pr != rows:
        t[rows:].zero_()
    return t

"""autograd.Function wiring of the HIP kernels: one Function per transformer sub-layer (pre-LN attention,
pre-LN MLP, pre-LN cross-attention), plus linear / layer-norm / token-assembly / position-table / loss ops.

Every forward and backward below is a fixed sequence of C-ABI kernel launches (afft_amd.ops); torch is
only the allocator, the stream and the autograd tape.  Residual stream, LayerNorm statistics, softmax
probabilities, losses, master weights and gradients are fp32; GEMM operands are bf16 (speed mode) or fp32
(parity mode) according to afft_amd.runtime.precision().
"""
from __future__ import annotations

import contextlib
import ctypes as C
import threading
from typing import NamedTuple, List, Optional, Tuple

import torch

from . import _lib as L_, ops, runtime as rt
from ._lib import (ACT_DGELU_ERF, ACT_DGELU_TANH, ACT_GELU_ERF, ACT_GELU_TANH, ACT_NONE, ACT_RELU, ACT_SIGMOID_GATE,
                   MASK_BLOCKCAUSAL, MASK_CAUSAL, MASK_DIAG, MASK_NONE)

Tensor = torch.Tensor


# --------------------------------------------------------------------------- activation buffers
class Act:
    """A GEMM operand/result buffer [rows, width] in the activation dtype. In bf16 mode it is allocated
    [pad64(rows), pad64(width)] with zero tails so it can be the k-contiguous operand of the next GEMM
    (K = pad64(width)) and the k-strided operand of a wgrad GEMM (K = pad64(rows))."""
    __slots__ = ("buf", "rows", "width", "_split", "_b16")

    def __init__(self, rows: int, width: int, device, dtype=None, like: Optional[Tensor] = None):
        dtype = dtype or rt.act_dtype()
        self.rows, self.width = rows, width
        self._split = None
        self._b16 = None
        if dtype == torch.bfloat16:
            pr, pw = rt.pad64(rows), rt.pad64(width)
            self.buf = (torch.empty if (pr == rows and pw == width) else torch.zeros)(pr, pw, dtype=dtype, device=device)
        else:
            self.buf = torch.empty(rows, width, dtype=dtype, device=device)

    @classmethod
    def carve(cls, flat: Tensor, offset: int, rows: int, width: int) -> "Act":
        """bf16 Act over flat[offset : offset + pad64(rows) * width] (width % 64 == 0); the row tail is zeroed by whoever
        fills it (the composite entry points do)."""
        a = cls.__new__(cls)
        a.rows, a.width, a._split, a._b16 = rows, width, None, None
        a.buf = flat[offset:offset + rt.pad64(rows) * width].view(rt.pad64(rows), width)
        return a

    @property
    def live(self) -> Tensor:            # [rows, width]
        return self.buf[:self.rows, :self.width]

    @property
    def k(self) -> Tensor:               # [rows, pad(width)] : A operand, k-contiguous
        return self.buf[:self.rows]

    @property
    def tn(self) -> Tensor:              # [pad(rows), width] : wgrad operand, k-strided
        return self.buf[:, :self.width]

    def cols(self, c0: int, c1: int) -> Tensor:
        return self.buf[:self.rows, c0:c1]

    def split(self) -> "ops.Split":
        """bf16x3 mode: the two-plane bf16 split of this (fp32) buffer, made once and used by every GEMM that reads it
        (forward / dgrad as the k-contiguous operand, wgrad as the k-strided one)."""
        f16 = rt.split_mode() == "f16"      # fp16x2: fp16 planes (forward only)
        if self._split is None or self._split.f16 != f16:
            self._split = ops.Split(self.live, f16=f16)
        return self._split


def _b16(act: Optional[Act]) -> Optional[Act]:
    """Backward pass of the 'fp16x2' precision (it runs as the bf16 mode, runtime.backward_precision): the bf16 operand copy of an
    activation the call-by-call forward saved in fp32 -- made once, on the main stream, on first use.  Anything else passes."""
    if act is None or act.buf.dtype != torch.float32 or rt.precision() != "bf16":
        return act
    if act._b16 is None:
        c = Act(act.rows, act.width, act.buf.device, dtype=torch.bfloat16)
        ops.cast(act.live, c.live)
        act._b16 = c
    return act._b16


def _in_backward_precision(fn):
    """decorator of a Function.backward: run it in the precision the backward pass of the current mode uses"""
    import functools

    @functools.wraps(fn)
    def wrapper(ctx, *grads):
        with rt.precision_scope(rt.backward_precision()):
            return fn(ctx, *grads)
    return wrapper


def to_act(x: Tensor, drop=None) -> Act:
    """fp32 [rows, width] (any row stride) -> Act in the activation dtype (a cast kernel in bf16 mode;
    in fp32 mode the tensor is used in place when dense).  drop: optional _lib.Dropout replayed/applied
    element-wise during the copy (backward of an epilogue dropout, or forward dropout of a GEMM input)."""
    rows, width = x.shape
    if rt.fp32_acts() and drop is None:
        a = Act.__new__(Act)
        a.rows, a.width = rows, width
        a.buf = x if x.stride(1) == 1 else x.contiguous()
        a._split = None
        a._b16 = None
        return a
    a = Act(rows, width, x.device)
    ops.cast(x, a.live, drop=drop)
    return a


# --------------------------------------------------------------------------- linear algebra helpers
def _lin_fwd(x: Act, W: Tensor, conv1d: bool, out: Tensor, **ep) -> Tensor:
    """out[rows, n_out] = epilogue(x @ W^T) (nn.Linear, W [out,in]) or x @ W (HF Conv1D, W [in,out])."""
    if rt.precision() == "bf16":
        w16 = rt.weight_images(W)            # [pad(rows), pad(cols)]
        if not conv1d:                       # W [out, in] is B stored [N, K]: "NT"
            pk = rt.weight_packed(W, x.k.shape[0]) if x.k.shape[1] == W.shape[1] else None
            return ops.gemm(x.k, w16[:W.shape[0]], out, b_t=True, b_packed=pk, **ep)
        return ops.gemm(x.k, w16[:, :W.shape[1]], out, **ep)             # W [in, out] = [K, N]: "NN"
    if rt.precision() == "fp16x2":           # x = hi + lo in fp16 planes, W rounded once to fp16 (its FP16 image)
        h16 = rt.weight_f16(W)
        if not conv1d:
            return ops.gemm(x.split(), h16[:W.shape[0]], out, b_t=True, **ep)
        return ops.gemm(x.split(), h16[:, :W.shape[1]], out, **ep)
    if rt.split_mode() is not None:          # bf16x3: three bf16 passes
        return ops.gemm(x.split(), rt.weight_split(W), out, b_t=not conv1d, **ep)
    return ops.gemm(x.live, W, out, b_t=not conv1d, **ep)


def _forward_only_check():
    if rt.precision() == "fp16x2":      # every Function.backward runs under _in_backward_precision: this would be a wiring error
        raise RuntimeError("afft_amd: a backward GEMM was reached in precision 'fp16x2' (a forward format: its backward pass runs "
                           "in runtime.backward_precision())")


def _lin_dgrad(dy: Act, W: Tensor, conv1d: bool, out: Tensor, **ep) -> Tensor:
    """out[rows, n_in] = epilogue(dy @ W) (nn.Linear) or dy @ W^T (Conv1D)."""
    if rt.precision() == "bf16":
        w16 = rt.weight_images(W)
        if conv1d:                           # dx = dy W^T, W [in, out] is B stored [N, K]: "NT"
            return ops.gemm(dy.k, w16[:W.shape[0]], out, b_t=True, **ep)
        return ops.gemm(dy.k, w16[:, :W.shape[1]], out, **ep)            # W [out, in] = [K, N]: "NN"
    _forward_only_check()
    if rt.precision() == "bf16x3":
        return ops.gemm(dy.split(), rt.weight_split(W), out, b_t=conv1d, **ep)
    return ops.gemm(dy.live, W, out, b_t=conv1d, **ep)


class _ThreadState(threading.local):
    """Everything this module remembers between two calls lives here, per THREAD: a forward pass runs on one thread per
    device (the main thread; nn.DataParallel's replica threads, test.py:130 of the reference) and a backward pass on the
    autograd engine's one worker thread per device, and every hand-off below (noting a sub-layer's output and looking it
    up, emitting a shadow and picking it up, queueing readiness notifications and flushing them, opening and closing a
    side-stream block) is between two calls made by the same thread for the same device."""

    def __init__(self):
        self.side_stream = None      # the auxiliary stream while a `with _Side(dev):` block is open
        self.main_stream = None      # ... and the stream the block was entered from
        self.join_queued = False     # a join of the auxiliary stream is queued for the end of the running backward
        self.pending_ready = []      # parameters whose gradient was enqueued by the running Function.backward
        self.last_out = None         # _Up of the sub-layer that ran last in this thread's forward
        self.shadow = None           # the hand-over emitted by the LayerNorm backward that ran last in this thread
        with _ALL_LOCK:     # nn.DataParallel starts fresh replica threads every forward: drop the finished ones
            _ALL_STATES[:] = [(t, d) for t, d in _ALL_STATES if t.is_alive()]
            _ALL_STATES.append((threading.current_thread(), self.__dict__))


_ALL_LOCK = threading.Lock()
_ALL_STATES: list = []     # (thread, its state dictionary): flush_ready(all_threads=True) walks them
_TS = _ThreadState()


def _on_side(t: Tensor) -> Tensor:
    """The tensor is about to be read by a kernel on the auxiliary stream: tell the caching allocator, so that its
    memory is not handed out again (to main-stream allocations) before that kernel has run."""
    if _TS.side_stream is not None and t.is_cuda:
        if rt.CAPTURING:
            rt.KEEPALIVE.append(t)
        else:
            t.record_stream(_TS.side_stream)
    return t


def _split_for_side(act: Act):
    """bf16x3: the operand split of `act` for a GEMM on the auxiliary stream.  A split is cached on the Act and later read
    by main-stream GEMMs too (the dgrad that follows the side block), so it is always MADE on the main stream; the
    auxiliary stream then waits for it."""
    if act._split is None and _TS.side_stream is not None:
        with torch.cuda.stream(_TS.main_stream):
            act.split()
            ev = torch.cuda.Event()
            ev.record()
        _TS.side_stream.wait_event(ev)
    return act.split()


def _wgrad(dy: Act, x: Act, W: Tensor, conv1d: bool) -> Optional[Tensor]:
    """dW = dy^T x (nn.Linear) or x^T dy (Conv1D); accumulated into W.grad (sink mode) or returned."""
    a, b = (x, dy) if conv1d else (dy, x)
    _forward_only_check()
    if rt.precision() == "bf16x3":
        at, bt = _split_for_side(a), _split_for_side(b)
        _on_side(at.planes)
        _on_side(bt.planes)
    else:
        at, bt = a.tn, b.tn
        _on_side(a.buf)
        _on_side(b.buf)
    if rt.grad_mode() == "sink":
        g, acc = rt.SINK.grad_buffer(W)
        ops.gemm(at, bt, g, a_t=True, accumulate=acc)
        _ready(W)
        return None
    g = torch.empty_like(W)
    ops.gemm(at, bt, g, a_t=True)
    return g


def _bgrad(dy: Tensor, bias: Optional[Tensor]) -> Optional[Tensor]:
    if bias is None:
        return None
    _on_side(dy)
    if rt.grad_mode() == "sink":
        g, acc = rt.SINK.grad_buffer(bias)
        ops.colsum(dy, g, accumulate=acc)
        _ready(bias)
        return None
    g = torch.empty_like(bias)
    ops.colsum(dy, g, accumulate=False)
    return g


class _Side:
    """`with _Side(dev):` enqueues the enclosed kernels on the device's auxiliary stream, ordered after everything
    already on the current stream.  Weight-gradient GEMMs (and bias column sums) are independent of the
    data-gradient chain of the same backward, so running them there lets their workgroups fill the CUs that a
    partial wave of the dgrad kernel leaves idle (256x256 tiles run one workgroup per CU).  Nothing on the main
    stream waits for them until the whole backward pass is over (join_side): their inputs are kept away from the
    allocator by record_stream (_on_side), and the optimizer / all-reduce stream orders itself behind the auxiliary
    stream when it picks up a bucket (parallel.GradReducer._launch)."""

    def __init__(self, device):
        self.on = rt.overlap_wgrad() and device.type == "cuda"
        self.device = device

    def __enter__(self):
        if self.on:
            ev = torch.cuda.Event()
            ev.record()
            st = rt.aux_stream(self.device)
            st.wait_event(ev)
            _TS.main_stream = torch.cuda.current_stream()
            self.ctx = torch.cuda.stream(st)
            self.ctx.__enter__()
            _TS.side_stream = st
        return self

    def __exit__(self, *exc):
        if self.on:
            _TS.side_stream = None
            self.ctx.__exit__(*exc)
        return False


def join_side(device):
    """Order the current stream after the auxiliary stream.  Sink mode: once, when the running backward pass ends
    (an autograd-engine callback) -- a per-sub-layer join would stall the dgrad chain behind every weight gradient
    and costs a cross-stream bubble each time.  Autograd mode: right away (the gradients returned to autograd are
    consumed on the current stream)."""
    if not (rt.overlap_wgrad() and device.type == "cuda"):
        return
    main, aux = torch.cuda.current_stream(), rt.aux_stream(device)
    if rt.grad_mode() == "sink":
        if not _TS.join_queued:
            ts = _TS.__dict__

            def _final(main=main, aux=aux, ts=ts):      # runs on whichever thread ends the backward pass
                ts["join_queued"] = False
                main.wait_stream(aux)
            try:
                torch.autograd.Variable._execution_engine.queue_callback(_final)
                _TS.join_queued = True
                return
            except RuntimeError:      # not inside a backward pass (a Function.backward called by hand)
                pass
        else:
            return
    main.wait_stream(aux)


def _ready(p: Tensor):
    """The gradient of p has been enqueued.  Notification is deferred to the end of the running backward
    (flush_ready) so that a consumer -- bucket all-reduce, per-bucket SGD on a side stream -- is ordered after
    every kernel of this backward that still READS the parameter (e.g. the dgrad GEMM reads the weight image the
    optimizer is about to overwrite)."""
    if rt.SINK.on_grad_ready is not None:
        _TS.pending_ready.append(p)


def flush_ready(all_threads: bool = False):
    """Deliver the queued notifications of this thread (all_threads: of every thread -- GradReducer.finish_step, after
    backward() has returned, so no worker is appending any more)."""
    cb = rt.SINK.on_grad_ready
    if all_threads:
        with _ALL_LOCK:
            lists = [d["pending_ready"] for _, d in _ALL_STATES]
    else:
        lists = [_TS.pending_ready]
    for pending in lists:
        if cb is not None:
            for p in pending:
                cb(p)
        pending.clear()


def settle_joins(device=None, drop_pending: bool = False):
    """After backward() has returned (or was abandoned by an exception): a join of the auxiliary stream that join_side queued for
    the end of the backward pass but that never ran -- the engine drops its callbacks when a Function raises -- is carried out
    now, and the flag that says "a join is queued" is cleared on every thread.  Without this the flag outlives the failed pass
    and every later backward pass on that thread would skip its join (the next forward pass would then read weights the
    auxiliary stream is still updating)."""
    stuck = False
    with _ALL_LOCK:
        for _, d in _ALL_STATES:
            if d.get("join_queued"):
                d["join_queued"] = False
                stuck = True
            if drop_pending:      # a step is about to start: readiness notes left over from an abandoned pass belong to nobody
                d["pending_ready"].clear()
    if stuck and torch.cuda.is_available() and rt.overlap_wgrad():
        dev = device if device is not None and device.type == "cuda" else torch.device("cuda", torch.cuda.current_device())
        torch.cuda.current_stream(dev).wait_stream(rt.aux_stream(dev))
    return stuck


# --------------------------------------------------------------------------- gradient hand-over between sub-layers
# The residual-stream gradient dx a sub-layer's backward returns is the dy of the sub-layer that ran before it in the
# forward pass, whose first two steps are "cast dy to bf16 with my output-dropout mask replayed" and "column-sum that
# for my output bias".  The LayerNorm-backward kernel that writes dx can emit both on the way (it has the row in
# registers), which saves a read of dy, a cast kernel and a column-sum kernel per sub-layer.  Who that upstream
# sub-layer is, is noted in the forward pass (_note_output / _upstream_of, matched on the tensor the two share); the
# emitted pieces travel in a single-slot hand-over (the thread's `shadow`) that the upstream backward accepts only for the very
# tensor (storage, shape, version counter) it was made from -- anything else (summed gradients, hooks, a different
# graph) falls back to the separate kernels.
class _Up(NamedTuple):
    ptr: int
    shape: tuple
    od: object          # _lib.Dropout or None: the upstream sub-layer's output dropout
    bias: object        # its output bias parameter (or None)


class _Shadow(NamedTuple):
    ptr: int
    shape: tuple
    version: int
    od: object
    bias: object
    act: Act
    gbias: object       # autograd mode: the bias gradient tensor (returned to autograd on acceptance)
    sink_direct: bool   # sink mode: the kernel wrote the column sums straight into bias.grad as the FIRST touch of the step
    sink_tmp: object    # sink mode, bias already touched this step: the column sums wait here and are added on acceptance


def _same_drop(a, b) -> bool:
    if a is None or b is None:
        return a is None and b is None
    return (a.p, a.key, a.path_p, a.path_key, a.path_group) == (b.p, b.key, b.path_p, b.path_key, b.path_group)


def _note_output(y: Tensor, od, bias):
    ok = rt.precision() == "bf16" or rt.backward_precision() == "bf16"      # the hand-over is a bf16 operand of the BACKWARD pass
    _TS.last_out = _Up(y.data_ptr(), tuple(y.shape), od, bias) if ok else None


def _upstream_of(x: Tensor) -> Optional[_Up]:
    u = _TS.last_out
    if u is not None and u.ptr == x.data_ptr() and u.shape == tuple(x.shape) and rt.handover():
        return u
    return None


def _discard(sh: Optional[_Shadow]):
    """A hand-over nobody accepted: the bias gradient it carries must not count.  Written straight into bias.grad as the
    first touch of the step -> forget the touch, so that whoever produces the real gradient overwrites it (and
    GradSink.finish_step zeroes it if nobody does); parked in a scratch vector -> just dropped.  Readiness is only ever
    notified on acceptance (_accept_bias), so the bucket counts do not depend on which way a hand-over went."""
    if sh is not None and sh.sink_direct and sh.bias is not None:
        rt.SINK.touched[id(sh.bias)] = False


def _take_shadow(dy: Tensor, od, bias) -> Optional[_Shadow]:
    sh, _TS.shadow = _TS.shadow, None
    if (sh is not None and sh.ptr == dy.data_ptr() and sh.shape == tuple(dy.shape) and sh.version == dy._version
            and sh.bias is bias and _same_drop(sh.od, od)):
        return sh
    _discard(sh)
    return None


def _accept_bias(sh: _Shadow):
    """The accepted hand-over's output-bias gradient: autograd mode returns the tensor; sink mode commits it now."""
    if sh.bias is None:
        return None
    if rt.grad_mode() != "sink":
        return sh.gbias
    if sh.sink_tmp is not None:
        g, acc = rt.SINK.grad_buffer(sh.bias)
        # the buffer was already touched this step, possibly by a kernel still pending on the auxiliary stream (a shared
        # sub-layer used twice): commit there, behind it and behind the LayerNorm backward that wrote the scratch vector
        side = _TS.side_stream
        own = side is None and g.is_cuda and rt.overlap_wgrad()
        if own:
            side = rt.aux_stream(g.device)
            side.wait_stream(torch.cuda.current_stream())
        ctx = torch.cuda.stream(side) if own else contextlib.nullcontext()
        with ctx:
            if acc:
                g.add_(sh.sink_tmp)
            else:
                g.copy_(sh.sink_tmp)
        if side is not None and g.is_cuda:
            if rt.CAPTURING:
                rt.KEEPALIVE.append(sh.sink_tmp)
            else:
                sh.sink_tmp.record_stream(side)
    _ready(sh.bias)
    return None


def _drop_shadow():
    sh, _TS.shadow = _TS.shadow, None
    _discard(sh)


def _forget_output():
    """Called by the forward of every op that is not a sub-layer: the next sub-layer has no sub-layer upstream."""
    _TS.last_out = None


def _ln_bwd(dy: Tensor, x: Tensor, w: Optional[Tensor], b: Optional[Tensor], mean, rstd, dx_in: Optional[Tensor],
            dx_out: Optional[Tensor] = None, up: Optional[_Up] = None):
    """returns (dx, dw, db) with dw/db None in sink mode (accumulated in place).  `up`: the sub-layer that consumes dx
    (see above); its bf16 operand and bias gradient are emitted by the same kernel and left in this thread's shadow slot."""
    rows, d = x.shape
    if dx_out is None:
        dx_out = torch.empty(rows, d, dtype=torch.float32, device=x.device)
    sink = rt.grad_mode() == "sink"
    extra = {}
    dxa = gbu = None
    direct, tmp = False, None
    if up is not None and d % 64 == 0 and dx_out.stride(0) == d:
        _drop_shadow()      # an earlier hand-over that was never picked up
        dxa = Act(rows, d, x.device)
        extra = dict(dx_bf16=dxa.live, copy_drop=up.od)
        if up.bias is not None:
            # the upstream sub-layer may still turn the hand-over down (autograd summed other gradients into dy, hooks,
            # another graph): nothing is committed here that _discard cannot take back
            if sink and up.bias.grad is not None and not rt.SINK.touched.get(id(up.bias), False):
                gbu, direct = up.bias.grad, True
                rt.SINK.touched[id(up.bias)] = True
            else:
                gbu = tmp = torch.empty_like(up.bias)
            extra.update(dcol=gbu, dcol_accumulate=False)
    if sink:
        gw = gb = None
        acc_w = acc_b = True
        if w is not None:
            gw, acc_w = rt.SINK.grad_buffer(w)
        if b is not None:
            gb, acc_b = rt.SINK.grad_buffer(b)
        if w is not None and b is not None and acc_w != acc_b:   # mixed first-touch state: fall back to a zero fill
            (gw if not acc_w else gb).zero_()
            acc_w = acc_b = True
        ops.layernorm_bwd(dy, x, w, mean, rstd, dx_out, dx_in=dx_in, dw=gw, db=gb,
                          accumulate=acc_w if w is not None else acc_b, **extra)
        if w is not None:
            _ready(w)
        if b is not None:
            _ready(b)
        gw = gb = None
    else:
        gw = torch.empty_like(w) if w is not None else None
        gb = torch.empty_like(b) if b is not None else None
        ops.layernorm_bwd(dy, x, w, mean, rstd, dx_out, dx_in=dx_in, dw=gw, db=gb, accumulate=False, **extra)
    if dxa is not None:
        _TS.shadow = _Shadow(dx_out.data_ptr(), tuple(dx_out.shape), dx_out._version, up.od, up.bias, dxa,
                          None if sink else gbu, direct, tmp if sink else None)
    return dx_out, gw, gb


def _stats(rows, device):
    return torch.empty(rows, dtype=torch.float32, device=device), torch.empty(rows, dtype=torch.float32, device=device)


def _attn_drop(drop):
    return (drop.p_attn, drop.k_attn) if drop is not None else (0.0, 0)


def _out_drop(drop):
    return drop.out_desc() if drop is not None else None


_GELU = {"erf": (ACT_GELU_ERF, ACT_DGELU_ERF), "tanh": (ACT_GELU_TANH, ACT_DGELU_TANH)}
_MASK = {"none": MASK_NONE, "diag": MASK_DIAG, "causal": MASK_CAUSAL, "blockcausal": MASK_BLOCKCAUSAL}


def _mask_args(mask):
    """'none' | 'diag' | 'causal' | ('blockcausal', T)  ->  (kernel mask id, period)"""
    if isinstance(mask, str):
        return _MASK[mask], 0
    return _MASK[mask[0]], int(mask[1])


def mask_table(mask) -> Optional[Tensor]:
    """('table', fp32 [L, L] device tensor): an arbitrary additive mask (models/transformerblock.py:26-28) -- the attention core then runs on
    the generic kernel with the table added to the scores (afft_attention_fwd_table), call by call (no composite entry point takes a table)"""
    return mask[1] if isinstance(mask, tuple) and mask[0] == "table" else None


def _attention_fwd(q, k, v, nseq, L, H, hd, scale, mask, out, probs, drop):
    tab = mask_table(mask)
    if tab is not None:
        return ops.attention_fwd_table(q, k, v, nseq, L, H, hd, scale, tab, out, probs, *(_attn_drop(drop)))
    mk, per = _mask_args(mask)
    return ops.attention_fwd(q, k, v, nseq, L, H, hd, scale, mk, out, probs, *(_attn_drop(drop)), mask_period=per)


# --------------------------------------------------------------------------- composite path: one C-ABI call per sub-layer
# afft_{attn,mlp,cross_attn}_sublayer_{fwd,bwd} (include/afft_hip.h, csrc/sublayer.hip) enqueue exactly the kernel sequences
# written out call by call in the three Functions below.  What stays here is bookkeeping: buffers (one allocation for the
# saved activations, one for the backward scratch), gradient-sink state, the hand-over, readiness notifications.
def _composite_ok(x: Tensor, pre_ln: bool, *widths: int, f16x2: bool = True) -> bool:
    """f16x2: the sub-layer has an fp16 two-pass forward (self-attention and MLP; the cross-attention composite is bf16 only)"""
    return (rt.composite() and pre_ln and rt.precision() in (("bf16", "fp16x2") if f16x2 else ("bf16",)) and x.is_cuda
            and x.stride(0) == x.shape[1] and all(w % 64 == 0 for w in widths))


def attn_take_ok(x: Tensor, L: int, H: int, pre_ln: bool = True) -> bool:
    """AttnSublayer(take=L) -- the output projection on token 0 of every sequence only -- exists on the composite path, for row
    counts whose quotient by L is a multiple of 64 (the weight-gradient GEMM reduces over whole 64-row K-tiles of the strided rows)"""
    R, d = x.shape
    hd = d // H
    return (L > 1 and R % L == 0 and (R // L) % 64 == 0 and _composite_ok(x, pre_ln, d)
            and (rt.precision() != "fp16x2" or (L <= 64 and hd % 64 == 0 and hd <= 1024)))


def _lo8_ok(conv1d: bool, R: int, *shapes) -> bool:
    """the fp8 lo pass for a sub-layer: nn.Linear weights and every (N, K) of its GEMMs on the 256x256 kernel (afft_gemm_lo8_ok)"""
    if conv1d or not rt.lo8():
        return False
    lib = L_.lib()
    return all(lib.afft_gemm_lo8_ok(int(R), int(n), int(k)) for n, k in shapes)


def _img_h(W: Tensor):
    """(pointer, leading dimension) of the FP16 image of a 2-D weight ('fp16x2' forward)"""
    h = rt.weight_f16(W)
    return h.data_ptr(), h.stride(0)


def _img(W: Tensor):
    """(pointer, leading dimension) of the bf16 image of a 2-D weight"""
    w16 = rt.weight_images(W)
    return w16.data_ptr(), w16.stride(0)


def _pk(W: Tensor, conv1d: bool, rows: int):
    """pointer of the fragment-packed copy of an nn.Linear weight's image (afft_gemm_t.b_packed) for a GEMM over `rows` rows, or None"""
    if conv1d:
        return None
    pk = rt.weight_packed(W, rows)
    return None if pk is None else pk.data_ptr()


def _ptr(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


def _grad_slot(p: Optional[Tensor], fresh: list):
    """(gradient tensor or None, accumulate flag) for parameter p: the sink's buffer, or a fresh tensor (autograd mode)"""
    if p is None:
        return None, 0
    if rt.grad_mode() == "sink":
        g, acc = rt.SINK.grad_buffer(p)
        return g, int(acc)
    g = torch.empty_like(p)
    fresh.append(g)
    return g, 0


def _ln_grad_slots(w: Optional[Tensor], b: Optional[Tensor], fresh: list):
    """LayerNorm weight / bias gradient buffers with ONE accumulate flag (see _ln_bwd for the mixed first-touch case)"""
    gw, aw = _grad_slot(w, fresh)
    gb, ab = _grad_slot(b, fresh)
    if w is not None and b is not None and aw != ab:
        (gw if not aw else gb).zero_()
        aw = ab = 1
    return gw, gb, (aw if w is not None else ab)


class _HandOut(NamedTuple):
    dxa: Optional[Act]
    gbu: Optional[Tensor]
    direct: bool
    tmp: Optional[Tensor]


def _plan_handover(up: Optional[_Up], rows: int, d: int, dev) -> _HandOut:
    """What the LayerNorm backward at the end of a composite backward emits for the sub-layer upstream (cf. _ln_bwd)"""
    if up is None or d % 64 != 0:
        return _HandOut(None, None, False, None)
    _drop_shadow()
    dxa = Act(rows, d, dev)
    gbu, direct, tmp = None, False, None
    if up.bias is not None:
        if rt.grad_mode() == "sink" and up.bias.grad is not None and not rt.SINK.touched.get(id(up.bias), False):
            gbu, direct = up.bias.grad, True
            rt.SINK.touched[id(up.bias)] = True
        else:
            gbu = tmp = torch.empty_like(up.bias)
    return _HandOut(dxa, gbu, direct, tmp)


def _publish_handover(ho: _HandOut, up: Optional[_Up], dx: Tensor):
    if ho.dxa is not None:
        sink = rt.grad_mode() == "sink"
        _TS.shadow = _Shadow(dx.data_ptr(), tuple(dx.shape), dx._version, up.od, up.bias, ho.dxa,
                             None if sink else ho.gbu, ho.direct, ho.tmp if sink else None)


def _streams(dev, rows: int = 1 << 30):
    """(raw main stream, raw auxiliary stream or None, torch auxiliary stream or None)"""
    main = ops._stream()
    if rt.overlap_wgrad():
        aux = rt.aux_stream(dev)
        return main, aux.cuda_stream, aux
    return main, None, None


def _fill_ws(s, dev, main_raw, aux_raw):
    ws = ops.gemm_workspace(dev, main_raw)
    s.gemm_ws, s.gemm_ws_bytes = ws.data_ptr(), ws.numel()
    if aux_raw is not None:
        wa = ops.gemm_workspace(dev, aux_raw)
        s.gemm_ws_aux, s.gemm_ws_aux_bytes = wa.data_ptr(), wa.numel()


def _keep_for_aux(aux, *tensors):
    """tensors a kernel on the auxiliary stream reads or writes: keep their memory from the allocator until it has run"""
    if aux is None:
        return
    for t in tensors:
        if t is not None and t.is_cuda:
            if rt.CAPTURING:
                rt.KEEPALIVE.append(t)
            else:
                t.record_stream(aux)


def _fuse_updates(s, weights) -> list:
    """weights: (struct field, parameter, accumulate flag).  For every weight the Trainer updates in the epilogue of its
    gradient GEMM (runtime.GradSink.fused_desc) point the field at its afft_sgd_fused_t; returns what must outlive the call."""
    keep = []
    for field, w, acc in weights:
        d = rt.SINK.fused_desc(w) if rt.grad_mode() == "sink" else None
        if d is not None and not acc:
            setattr(s, field, C.pointer(d))
            keep.append(d)
            rt.SINK.fused_applied[id(w)] = rt.SINK.fused_applied.get(id(w), 0) + 1     # audited by Trainer._audit_fused_step
    return keep


_TAP = None      # diagnostics (tools/repro_diag.py): called with (kind, tensors) right after a composite backward has been enqueued


def _ln_partial(rows: int, d: int, dev) -> Tensor:
    return torch.empty(L_.lib().afft_layernorm_bwd_nparts(rows) * 3 * d, dtype=torch.float32, device=dev)


def _attn_fwd_c(ctx, x, ln_w, ln_b, w_qkv, b_qkv, w_proj, b_proj, L, H, mask, eps, conv1d, scale, drop, probs_out=None, take=0):
    R, d = x.shape
    dev = x.device
    nseq, pr = R // L, rt.pad64(R)
    Ry = R // take if take else R        # rows that leave the sub-layer
    ctx.up = _upstream_of(x)
    f16x2 = rt.precision() == "fp16x2"
    grad = any(ctx.needs_input_grad)
    # the bf16 activations the backward pass reads; in the fp16x2 forward they are COPIES beside the fp16 operand planes (hi + lo
    # of xn | qkv | ao, forward-only scratch) and a forward nobody differentiates does not make them
    saved = torch.empty(pr * 5 * d, dtype=torch.bfloat16, device=dev) if (grad or not f16x2) else None
    planes = torch.empty(2 * pr * 5 * d, dtype=torch.float16, device=dev) if f16x2 else None
    if saved is not None:
        xn, qkv, ao = Act.carve(saved, 0, R, d), Act.carve(saved, pr * d, R, 3 * d), Act.carve(saved, pr * 4 * d, R, d)
    else:
        xn = qkv = ao = None
    stats = torch.empty(2, R, dtype=torch.float32, device=dev)
    probs = probs_out if probs_out is not None else torch.empty(nseq, H, L, L, dtype=torch.float32, device=dev)
    y = torch.empty(Ry, d, dtype=torch.float32, device=dev)
    scale = float(scale) if scale else float(d // H) ** -0.5
    mk, per = _mask_args(mask)
    s = L_.AttnSublayer()
    s.rows, s.d, s.L, s.H, s.conv1d, s.mask, s.mask_period, s.eps, s.scale = R, d, L, H, int(conv1d), mk, per, eps, scale
    s.take = int(take)
    s.x, s.ln_w, s.ln_b = x.data_ptr(), _ptr(ln_w), _ptr(ln_b)
    if f16x2:
        s.f16x2 = 1
        s.w_qkv, s.ldw_qkv = _img_h(w_qkv)
        s.w_proj, s.ldw_proj = _img_h(w_proj)
        if _lo8_ok(conv1d, R, (3 * d, d)) and _lo8_ok(conv1d, Ry, (d, d)):      # second pass on the block-scaled fp8 MFMA: e4m3 lo planes of xn / ao, e4m3 weight images
            s.f16x2 = 2
            s.w_qkv8, s.w_proj8 = rt.weight_f8(w_qkv).data_ptr(), rt.weight_f8(w_proj).data_ptr()
        s.f16x2 |= rt.one_pass_flags(conv1d, d, "qkv", "proj", "attn")
        base = planes.data_ptr()
        s.xn, s.qkv, s.ao = base, base + 2 * (2 * pr * d), base + 2 * (2 * pr * 4 * d)      # [hi | lo] of xn, then of qkv, then of ao
        if saved is not None:
            s.xn_b, s.qkv_b, s.ao_b = xn.buf.data_ptr(), qkv.buf.data_ptr(), ao.buf.data_ptr()
    else:
        s.w_qkv, s.ldw_qkv = _img(w_qkv)
        s.w_proj, s.ldw_proj = _img(w_proj)
        s.w_qkv_pk, s.w_proj_pk = _pk(w_qkv, conv1d, R), _pk(w_proj, conv1d, Ry)
        s.xn, s.qkv, s.ao = xn.buf.data_ptr(), qkv.buf.data_ptr(), ao.buf.data_ptr()
    s.b_qkv, s.b_proj = _ptr(b_qkv), _ptr(b_proj)
    s.p_attn, s.k_attn = _attn_drop(drop)
    od = _out_drop(drop)
    if od is not None:
        s.out_drop = od
    s.mean, s.rstd, s.probs, s.y = stats[0].data_ptr(), stats[1].data_ptr(), probs.data_ptr(), y.data_ptr()
    main_raw = ops._stream()
    _fill_ws(s, dev, main_raw, None)
    L_.check(L_.lib().afft_attn_sublayer_fwd(C.byref(s), main_raw), "attn_sublayer_fwd")
    ctx.save_for_backward(x, ln_w, ln_b, w_qkv, b_qkv, w_proj, b_proj, stats[0], stats[1], probs)
    ctx.acts = (xn, qkv, ao, saved)
    ctx.cfg = (L, H, scale, conv1d, True, drop)
    ctx.mask = (mk, per)
    ctx.take = int(take)
    ctx.composite = True
    ctx.mark_non_differentiable(probs)
    _note_output(y, od, b_proj)
    return y, probs


def _attn_bwd_c(ctx, dy):
    x, ln_w, ln_b, w_qkv, b_qkv, w_proj, b_proj, mean, rstd, probs = ctx.saved_tensors
    xn, qkv, ao, saved = ctx.acts
    L, H, scale, conv1d, _, drop = ctx.cfg
    R, d = x.shape
    dev = x.device
    pr = rt.pad64(R)
    dy = dy.contiguous()
    od = _out_drop(drop)
    sh = _take_shadow(dy, od, b_proj)
    main_raw, aux_raw, aux = _streams(dev, R)
    fresh: list = []
    scratch = torch.empty(pr * 6 * d, dtype=torch.bfloat16, device=dev)      # dya | dao | dqkv | dxn
    s = L_.AttnSublayer()
    s.rows, s.d, s.L, s.H, s.conv1d, s.eps, s.scale = R, d, L, H, int(conv1d), 0.0, scale
    s.take = ctx.take
    s.mask, s.mask_period = ctx.mask
    s.x, s.ln_w = x.data_ptr(), _ptr(ln_w)
    s.w_qkv, s.ldw_qkv = _img(w_qkv)
    s.w_proj, s.ldw_proj = _img(w_proj)
    s.p_attn, s.k_attn = _attn_drop(drop)
    if od is not None:
        s.out_drop = od
    s.xn, s.qkv, s.ao = xn.buf.data_ptr(), qkv.buf.data_ptr(), ao.buf.data_ptr()
    s.mean, s.rstd, s.probs = mean.data_ptr(), rstd.data_ptr(), probs.data_ptr()
    s.dy = dy.data_ptr()
    base = scratch.data_ptr()
    if sh is not None:
        s.dya, s.dya_ready = sh.act.buf.data_ptr(), 1
        g_bp = _accept_bias(sh)
    else:
        s.dya, s.dya_ready = base, 0
        gb, acc = _grad_slot(b_proj, fresh)
        s.g_b_proj, s.acc_b_proj = _ptr(gb), acc
        g_bp = gb if rt.grad_mode() != "sink" else None
    s.dao, s.dqkv, s.dxn = base + pr * d * 2, base + pr * 2 * d * 2, base + pr * 5 * d * 2
    g_wq, s.acc_w_qkv = _grad_slot(w_qkv, fresh)
    g_bq, s.acc_b_qkv = _grad_slot(b_qkv, fresh)
    g_wp, s.acc_w_proj = _grad_slot(w_proj, fresh)
    g_lw, g_lb, s.acc_ln = _ln_grad_slots(ln_w, ln_b, fresh)
    s.g_w_qkv, s.g_b_qkv, s.g_w_proj, s.g_ln_w, s.g_ln_b = _ptr(g_wq), _ptr(g_bq), _ptr(g_wp), _ptr(g_lw), _ptr(g_lb)
    keep = _fuse_updates(s, (("sgd_w_qkv", w_qkv, s.acc_w_qkv), ("sgd_w_proj", w_proj, s.acc_w_proj)))
    dx = torch.empty(R, d, dtype=torch.float32, device=dev)
    s.dx = dx.data_ptr()
    ho = _plan_handover(ctx.up, R, d, dev)
    if ho.dxa is not None:
        s.dx_bf16 = ho.dxa.buf.data_ptr()
        if ctx.up.od is not None:
            s.up_drop = C.pointer(ctx.up.od)
        s.up_dcol = _ptr(ho.gbu)
    partial = _ln_partial(R, d, dev)
    s.ln_partial = partial.data_ptr()
    _fill_ws(s, dev, main_raw, aux_raw)
    if _TAP is not None:
        _TAP("attn_bwd_pre", dict(R=R, d=d, w_qkv=w_qkv, w_proj=w_proj))
    L_.check(L_.lib().afft_attn_sublayer_bwd(C.byref(s), main_raw, aux_raw), "attn_sublayer_bwd")
    if _TAP is not None:
        _TAP("attn_bwd", dict(scratch=scratch, dy=dy, dx=dx, R=R, d=d, w_qkv=w_qkv, w_proj=w_proj, saved=saved, shadow=sh, partial=partial, x=x))
    _keep_for_aux(aux, saved, scratch, dy, sh.act.buf if sh is not None else None, *fresh)
    _publish_handover(ho, ctx.up, dx)
    if rt.grad_mode() == "sink":
        for p in (w_proj, None if sh is not None else b_proj, w_qkv, b_qkv, ln_w, ln_b):
            if p is not None:
                _ready(p)
        g_wq = g_bq = g_wp = g_bp = g_lw = g_lb = None
    join_side(dev)
    ctx.acts = None
    flush_ready()
    return dx, g_lw, g_lb, g_wq, g_bq, g_wp, g_bp, None, None, None, None, None, None, None, None, None


def _mlp_fwd_c(ctx, x, ln_w, ln_b, w1, b1, w2, b2, eps, gelu, conv1d, hidden, drop):
    R, d = x.shape
    dev = x.device
    pr = rt.pad64(R)
    ctx.up = _upstream_of(x)
    f16x2 = rt.precision() == "fp16x2"
    grad = any(ctx.needs_input_grad)
    saved = torch.empty(pr * (d + 2 * hidden), dtype=torch.bfloat16, device=dev) if (grad or not f16x2) else None      # cf. _attn_fwd_c
    planes = torch.empty(2 * pr * (d + hidden), dtype=torch.float16, device=dev) if f16x2 else None      # [hi | lo] of xn, then of h
    if saved is not None:
        xn, u, h = Act.carve(saved, 0, R, d), Act.carve(saved, pr * d, R, hidden), Act.carve(saved, pr * (d + hidden), R, hidden)
    else:
        xn = u = h = None
    stats = torch.empty(2, R, dtype=torch.float32, device=dev)
    y = torch.empty(R, d, dtype=torch.float32, device=dev)
    s = L_.MLPSublayer()
    s.rows, s.d, s.hidden, s.conv1d, s.gelu, s.eps = R, d, hidden, int(conv1d), _GELU[gelu][0], eps
    s.x, s.ln_w, s.ln_b = x.data_ptr(), _ptr(ln_w), _ptr(ln_b)
    # the pre-activation is only read by backward: a forward nobody will differentiate (no_grad) does not store it (84 MB at R = 5120)
    if f16x2:
        s.f16x2 = 1
        s.w1, s.ldw1 = _img_h(w1)
        s.w2, s.ldw2 = _img_h(w2)
        if _lo8_ok(conv1d, R, (hidden, d), (d, hidden)):
            s.f16x2 = 2
            s.w1_8, s.w2_8 = rt.weight_f8(w1).data_ptr(), rt.weight_f8(w2).data_ptr()
        s.f16x2 |= rt.one_pass_flags(conv1d, d, "fc1", "fc2")
        base = planes.data_ptr()
        s.xn, s.h = base, base + 2 * (2 * pr * d)
        if saved is not None:
            s.xn_b, s.u, s.h_b = xn.buf.data_ptr(), u.buf.data_ptr(), h.buf.data_ptr()
    else:
        s.w1, s.ldw1 = _img(w1)
        s.w2, s.ldw2 = _img(w2)
        s.w1_pk, s.w2_pk = _pk(w1, conv1d, R), _pk(w2, conv1d, R)
        s.xn, s.u, s.h = xn.buf.data_ptr(), (u.buf.data_ptr() if grad else None), h.buf.data_ptr()
    s.b1, s.b2 = _ptr(b1), _ptr(b2)
    od = _out_drop(drop)
    if od is not None:
        s.out_drop = od
    s.mean, s.rstd, s.y = stats[0].data_ptr(), stats[1].data_ptr(), y.data_ptr()
    main_raw = ops._stream()
    _fill_ws(s, dev, main_raw, None)
    L_.check(L_.lib().afft_mlp_sublayer_fwd(C.byref(s), main_raw), "mlp_sublayer_fwd")
    ctx.save_for_backward(x, ln_w, ln_b, w1, b1, w2, b2, stats[0], stats[1])
    ctx.acts = (xn, u, h, saved)
    ctx.cfg = (gelu, conv1d, hidden, True, drop)
    ctx.composite = True
    _note_output(y, od, b2)
    return y


def _mlp_bwd_c(ctx, dy):
    x, ln_w, ln_b, w1, b1, w2, b2, mean, rstd = ctx.saved_tensors
    xn, u, h, saved = ctx.acts
    gelu, conv1d, hidden, _, drop = ctx.cfg
    R, d = x.shape
    dev = x.device
    pr = rt.pad64(R)
    dy = dy.contiguous()
    od = _out_drop(drop)
    sh = _take_shadow(dy, od, b2)
    main_raw, aux_raw, aux = _streams(dev, R)
    fresh: list = []
    scratch = torch.empty(pr * (2 * d + hidden), dtype=torch.bfloat16, device=dev)    # dya | dxn | du
    s = L_.MLPSublayer()
    s.rows, s.d, s.hidden, s.conv1d, s.gelu = R, d, hidden, int(conv1d), _GELU[gelu][0]
    s.x, s.ln_w = x.data_ptr(), _ptr(ln_w)
    s.w1, s.ldw1 = _img(w1)
    s.w2, s.ldw2 = _img(w2)
    if od is not None:
        s.out_drop = od
    s.xn, s.u, s.h = xn.buf.data_ptr(), u.buf.data_ptr(), h.buf.data_ptr()
    s.mean, s.rstd = mean.data_ptr(), rstd.data_ptr()
    s.dy = dy.data_ptr()
    base = scratch.data_ptr()
    if sh is not None:
        s.dya, s.dya_ready = sh.act.buf.data_ptr(), 1
        g_b2 = _accept_bias(sh)
    else:
        s.dya, s.dya_ready = base, 0
        gb, acc = _grad_slot(b2, fresh)
        s.g_b2, s.acc_b2 = _ptr(gb), acc
        g_b2 = gb if rt.grad_mode() != "sink" else None
    s.dxn, s.du = base + pr * d * 2, base + pr * 2 * d * 2
    g_w1, s.acc_w1 = _grad_slot(w1, fresh)
    g_b1, s.acc_b1 = _grad_slot(b1, fresh)
    g_w2, s.acc_w2 = _grad_slot(w2, fresh)
    g_lw, g_lb, s.acc_ln = _ln_grad_slots(ln_w, ln_b, fresh)
    s.g_w1, s.g_b1, s.g_w2, s.g_ln_w, s.g_ln_b = _ptr(g_w1), _ptr(g_b1), _ptr(g_w2), _ptr(g_lw), _ptr(g_lb)
    keep = _fuse_updates(s, (("sgd_w1", w1, s.acc_w1), ("sgd_w2", w2, s.acc_w2)))
    dx = torch.empty(R, d, dtype=torch.float32, device=dev)
    s.dx = dx.data_ptr()
    ho = _plan_handover(ctx.up, R, d, dev)
    if ho.dxa is not None:
        s.dx_bf16 = ho.dxa.buf.data_ptr()
        if ctx.up.od is not None:
            s.up_drop = C.pointer(ctx.up.od)
        s.up_dcol = _ptr(ho.gbu)
    partial = _ln_partial(R, d, dev)
    s.ln_partial = partial.data_ptr()
    _fill_ws(s, dev, main_raw, aux_raw)
    L_.check(L_.lib().afft_mlp_sublayer_bwd(C.byref(s), main_raw, aux_raw), "mlp_sublayer_bwd")
    if _TAP is not None:
        _TAP("mlp_bwd", dict(scratch=scratch, dy=dy, dx=dx, R=R, d=d, hidden=hidden, w1=w1, w2=w2, saved=saved, shadow=sh, partial=partial, x=x))
    _keep_for_aux(aux, saved, scratch, dy, sh.act.buf if sh is not None else None, *fresh)
    _publish_handover(ho, ctx.up, dx)
    if rt.grad_mode() == "sink":
        for p in (w2, None if sh is not None else b2, w1, b1, ln_w, ln_b):
            if p is not None:
                _ready(p)
        g_w1 = g_b1 = g_w2 = g_b2 = g_lw = g_lb = None
    join_side(dev)
    ctx.acts = None
    flush_ready()
    return dx, g_lw, g_lb, g_w1, g_b1, g_w2, g_b2, None, None, None, None, None


def _cross_fwd_c(ctx, x, mem, nq_w, nq_b, nkv_w, nkv_b, w_q, w_k, w_v, w_proj, b_proj, L, H, mask, eps, scale, drop):
    R, d = x.shape
    dev = x.device
    nseq, pr = R // L, rt.pad64(R)
    ctx.up = _upstream_of(x)
    saved = torch.empty(pr * 6 * d, dtype=torch.bfloat16, device=dev)
    xq, mkv, q, k, v, ao = (Act.carve(saved, i * pr * d, R, d) for i in range(6))
    stats = torch.empty(4, R, dtype=torch.float32, device=dev)
    probs = torch.empty(nseq, H, L, L, dtype=torch.float32, device=dev)
    y = torch.empty(R, d, dtype=torch.float32, device=dev)
    scale = float(scale) if scale else float(d // H) ** -0.5
    mk, per = _mask_args(mask)
    s = L_.CrossAttnSublayer()
    s.rows, s.d, s.L, s.H, s.mask, s.mask_period, s.eps, s.scale = R, d, L, H, mk, per, eps, scale
    s.x, s.mem = x.data_ptr(), mem.data_ptr()
    s.nq_w, s.nq_b, s.nkv_w, s.nkv_b = _ptr(nq_w), _ptr(nq_b), _ptr(nkv_w), _ptr(nkv_b)
    imgs = [_img(w) for w in (w_q, w_k, w_v, w_proj)]
    assert len({ld for _, ld in imgs}) == 1
    (s.w_q, s.ldw), (s.w_k, _), (s.w_v, _), (s.w_proj, _) = imgs
    s.b_proj = _ptr(b_proj)
    s.p_attn, s.k_attn = _attn_drop(drop)
    od = _out_drop(drop)
    if od is not None:
        s.out_drop = od
    s.xq, s.mkv, s.q, s.k, s.v, s.ao = (a.buf.data_ptr() for a in (xq, mkv, q, k, v, ao))
    s.mean_q, s.rstd_q, s.mean_kv, s.rstd_kv = (stats[i].data_ptr() for i in range(4))
    s.probs, s.y = probs.data_ptr(), y.data_ptr()
    main_raw = ops._stream()
    _fill_ws(s, dev, main_raw, None)
    L_.check(L_.lib().afft_cross_attn_sublayer_fwd(C.byref(s), main_raw), "cross_attn_sublayer_fwd")
    ctx.save_for_backward(x, mem, nq_w, nq_b, nkv_w, nkv_b, w_q, w_k, w_v, w_proj, b_proj, stats[0], stats[1], stats[2], stats[3],
                          probs)
    ctx.acts = (xq, mkv, q, k, v, ao, saved)
    ctx.cfg = (L, H, scale, True, drop)
    ctx.mask = (mk, per)
    ctx.composite = True
    _note_output(y, od, b_proj)
    return y


def _cross_bwd_c(ctx, dy):
    (x, mem, nq_w, nq_b, nkv_w, nkv_b, w_q, w_k, w_v, w_proj, b_proj, mq, rq, mkm, rk, probs) = ctx.saved_tensors
    xq, mkv, q, k, v, ao, saved = ctx.acts
    L, H, scale, _, drop = ctx.cfg
    R, d = x.shape
    dev = x.device
    pr = rt.pad64(R)
    dy = dy.contiguous()
    od = _out_drop(drop)
    sh = _take_shadow(dy, od, b_proj)
    main_raw, aux_raw, aux = _streams(dev, R)
    fresh: list = []
    scratch = torch.empty(pr * 6 * d, dtype=torch.bfloat16, device=dev)       # dya | dao | dq | dk | dv | dxq
    dmkv = torch.empty(R, d, dtype=torch.float32, device=dev)
    s = L_.CrossAttnSublayer()
    s.rows, s.d, s.L, s.H, s.scale = R, d, L, H, scale
    s.mask, s.mask_period = ctx.mask
    s.x, s.mem, s.nq_w, s.nkv_w = x.data_ptr(), mem.data_ptr(), _ptr(nq_w), _ptr(nkv_w)
    (s.w_q, s.ldw), (s.w_k, _), (s.w_v, _), (s.w_proj, _) = [_img(w) for w in (w_q, w_k, w_v, w_proj)]
    s.p_attn, s.k_attn = _attn_drop(drop)
    if od is not None:
        s.out_drop = od
    s.xq, s.mkv, s.q, s.k, s.v, s.ao = (a.buf.data_ptr() for a in (xq, mkv, q, k, v, ao))
    s.mean_q, s.rstd_q, s.mean_kv, s.rstd_kv, s.probs = mq.data_ptr(), rq.data_ptr(), mkm.data_ptr(), rk.data_ptr(), probs.data_ptr()
    s.dy = dy.data_ptr()
    base = scratch.data_ptr()
    if sh is not None:
        s.dya, s.dya_ready = sh.act.buf.data_ptr(), 1
        g_bp = _accept_bias(sh)
    else:
        s.dya, s.dya_ready = base, 0
        gb, acc = _grad_slot(b_proj, fresh)
        s.g_b_proj, s.acc_b_proj = _ptr(gb), acc
        g_bp = gb if rt.grad_mode() != "sink" else None
    s.dao, s.dq, s.dk, s.dv, s.dxq = (base + i * pr * d * 2 for i in range(1, 6))
    s.dmkv = dmkv.data_ptr()
    g_q, s.acc_w_q = _grad_slot(w_q, fresh)
    g_k, s.acc_w_k = _grad_slot(w_k, fresh)
    g_v, s.acc_w_v = _grad_slot(w_v, fresh)
    g_wp, s.acc_w_proj = _grad_slot(w_proj, fresh)
    g_qw, g_qb, s.acc_nq = _ln_grad_slots(nq_w, nq_b, fresh)
    g_kw, g_kb, s.acc_nkv = _ln_grad_slots(nkv_w, nkv_b, fresh)
    s.g_w_q, s.g_w_k, s.g_w_v, s.g_w_proj = _ptr(g_q), _ptr(g_k), _ptr(g_v), _ptr(g_wp)
    keep = _fuse_updates(s, (("sgd_w_q", w_q, s.acc_w_q), ("sgd_w_k", w_k, s.acc_w_k), ("sgd_w_v", w_v, s.acc_w_v),
                             ("sgd_w_proj", w_proj, s.acc_w_proj)))
    s.g_nq_w, s.g_nq_b, s.g_nkv_w, s.g_nkv_b = _ptr(g_qw), _ptr(g_qb), _ptr(g_kw), _ptr(g_kb)
    dx = torch.empty(R, d, dtype=torch.float32, device=dev)
    dmem = torch.empty(R, d, dtype=torch.float32, device=dev)
    s.dx, s.dmem = dx.data_ptr(), dmem.data_ptr()
    ho = _plan_handover(ctx.up, R, d, dev)
    if ho.dxa is not None:
        s.dx_bf16 = ho.dxa.buf.data_ptr()
        if ctx.up.od is not None:
            s.up_drop = C.pointer(ctx.up.od)
        s.up_dcol = _ptr(ho.gbu)
    partial = _ln_partial(R, 2 * d, dev)
    s.ln_partial, s.ln_partial2 = partial.data_ptr(), partial.data_ptr() + partial.numel() * 2
    _fill_ws(s, dev, main_raw, aux_raw)
    L_.check(L_.lib().afft_cross_attn_sublayer_bwd(C.byref(s), main_raw, aux_raw), "cross_attn_sublayer_bwd")
    _keep_for_aux(aux, saved, scratch, dy, sh.act.buf if sh is not None else None, *fresh)
    _publish_handover(ho, ctx.up, dx)
    if rt.grad_mode() == "sink":
        for p in (w_proj, None if sh is not None else b_proj, w_q, w_k, w_v, nkv_w, nkv_b, nq_w, nq_b):
            if p is not None:
                _ready(p)
        g_q = g_k = g_v = g_wp = g_bp = g_qw = g_qb = g_kw = g_kb = None
    join_side(dev)
    ctx.acts = None
    flush_ready()
    return dx, dmem, g_qw, g_qb, g_kw, g_kb, g_q, g_k, g_v, g_wp, g_bp, None, None, None, None, None, None, None, None, None, None



# --------------------------------------------------------------------------- pre-LN self-attention sub-layer
class AttnSublayer(torch.autograd.Function):
    """y = x + proj(attn(split(qkv(LN(x)))))   -- Block / DecoderBlock self-attention half
    (models/transformerblock.py:131-133,158-159) and GPT2Block's attention half (HF modeling_gpt2.py).
    x: fp32 [nseq*L, d].  Returns (y fp32 [nseq*L, d], probs fp32 [nseq, H, L, L])."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, w_qkv, b_qkv, w_proj, b_proj, L, H, mask, eps, conv1d, pre_ln=True, scale=None,
                drop=None, probs_out=None, take=0):
        """probs_out: fp32 [nseq, H, L, L] to write the attention maps into (the caller's slice of ONE buffer for all its blocks:
        the SA-Fuser returns the stacked maps, models/fusion.py:144 -- no torch.stack copy of every block's maps per forward)
        take = L: y is [nseq, d], token 0 of every sequence (attn_take_ok says whether this call can do that)"""
        R, d = x.shape
        assert take in (0, L) and (not take or attn_take_ok(x, L, H, pre_ln)), "AttnSublayer: take needs the composite path (attn_take_ok)"
        nseq, hd = R // L, d // H
        dev = x.device
        ctx.composite = False
        if probs_out is not None:
            assert probs_out.shape == (nseq, H, L, L) and probs_out.dtype == torch.float32 and probs_out.is_contiguous()
        # probs is returned for the caller's attention maps and takes no gradient: without this autograd hands backward a
        # freshly ZERO-FILLED tensor of its shape for it on every call (a fill kernel per attention sub-layer and step)
        ctx.set_materialize_grads(False)
        # fp16x2: the attention core on hi + lo planes exists on the MFMA path only (L <= 64, head dimension a multiple of 64)
        if mask_table(mask) is None and _composite_ok(x, pre_ln, d) and (rt.precision() != "fp16x2" or (L <= 64 and hd % 64 == 0 and hd <= 1024)):
            return _attn_fwd_c(ctx, x, ln_w, ln_b, w_qkv, b_qkv, w_proj, b_proj, L, H, mask, eps, conv1d, scale, drop, probs_out, take)
        ctx.up = _upstream_of(x) if pre_ln else None
        mean, rstd = _stats(R if pre_ln else 0, dev)
        if pre_ln:
            xn = Act(R, d, dev)
            ops.layernorm_fwd(x, ln_w, ln_b, eps, xn.live, mean, rstd)
        else:  # bare Attention module (models/transformerblock.py:19-36): no norm, no residual
            xn = to_act(x)
        qkv = Act(R, 3 * d, dev)
        _lin_fwd(xn, w_qkv, conv1d, qkv.live, bias=b_qkv)
        ao = Act(R, d, dev)
        probs = probs_out if probs_out is not None else torch.empty(nseq, H, L, L, dtype=torch.float32, device=dev)
        scale = float(scale) if scale else float(hd) ** -0.5
        _attention_fwd(qkv.cols(0, d), qkv.cols(d, 2 * d), qkv.cols(2 * d, 3 * d), nseq, L, H, hd, scale, mask, ao.live, probs, drop)
        y = torch.empty(R, d, dtype=torch.float32, device=dev)
        _lin_fwd(ao, w_proj, conv1d, y, bias=b_proj, residual=x if pre_ln else None, drop=_out_drop(drop))
        ctx.save_for_backward(x, ln_w, ln_b, w_qkv, b_qkv, w_proj, b_proj, mean, rstd, probs)
        ctx.acts = (xn, qkv, ao)
        ctx.cfg = (L, H, scale, conv1d, pre_ln, drop)
        ctx.mark_non_differentiable(probs)
        _note_output(y, _out_drop(drop), b_proj)
        return y, probs

    @staticmethod
    @_in_backward_precision
    def backward(ctx, dy, _dprobs):
        if dy is None:          # (set_materialize_grads(False)) nobody used y
            _drop_shadow()      # a hand-over meant for this backward and queued notifications must not outlive it
            flush_ready()
            return (None,) * 17
        if ctx.composite:
            return _attn_bwd_c(ctx, dy) + (None,)
        x, ln_w, ln_b, w_qkv, b_qkv, w_proj, b_proj, mean, rstd, probs = ctx.saved_tensors
        xn, qkv, ao = (_b16(t) for t in ctx.acts)
        L, H, scale, conv1d, pre_ln, drop = ctx.cfg
        R, d = x.shape
        nseq, hd = R // L, d // H
        dev = x.device
        dy = dy.contiguous()
        od = _out_drop(drop)
        sh = _take_shadow(dy, od, b_proj)
        dya = sh.act if sh is not None else to_act(dy, od)
        with _Side(dev):
            g_wp = _wgrad(dya, ao, w_proj, conv1d)
            g_bp = _accept_bias(sh) if sh is not None else _bgrad(dy if od is None else dya.live, b_proj)
        dao = Act(R, d, dev)
        _lin_dgrad(dya, w_proj, conv1d, dao.live)
        dqkv = Act(R, 3 * d, dev)
        ops.attention_bwd(dao.live, qkv.cols(0, d), qkv.cols(d, 2 * d), qkv.cols(2 * d, 3 * d), probs, nseq, L, H, hd,
                          scale, dqkv.cols(0, d), dqkv.cols(d, 2 * d), dqkv.cols(2 * d, 3 * d), *(_attn_drop(drop)))
        with _Side(dev):
            g_wq = _wgrad(dqkv, xn, w_qkv, conv1d)
            g_bq = _bgrad(dqkv.live, b_qkv)
        if pre_ln:
            dxn = Act(R, d, dev)
            _lin_dgrad(dqkv, w_qkv, conv1d, dxn.live)
            dx, g_lw, g_lb = _ln_bwd(dxn.live, x, ln_w, ln_b, mean, rstd, dx_in=dy, up=ctx.up)
        else:
            dx = torch.empty(R, d, dtype=torch.float32, device=dev)
            _lin_dgrad(dqkv, w_qkv, conv1d, dx)
            g_lw = g_lb = None
        join_side(dev)
        ctx.acts = None
        flush_ready()
        return dx, g_lw, g_lb, g_wq, g_bq, g_wp, g_bp, None, None, None, None, None, None, None, None, None


# --------------------------------------------------------------------------- pre-LN MLP sub-layer
class MLPSublayer(torch.autograd.Function):
    """y = x + fc2(act(fc1(LN(x))))  -- models/transformerblock.py:84-89,134 (exact-erf GELU) and HF GPT2MLP (gelu_new)."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, w1, b1, w2, b2, eps, gelu, conv1d, pre_ln=True, drop=None):
        R, d = x.shape
        dev = x.device
        hidden = w1.shape[1] if conv1d else w1.shape[0]
        d_out = w2.shape[1] if conv1d else w2.shape[0]
        ctx.composite = False
        if d_out == d and _composite_ok(x, pre_ln, d, hidden):
            return _mlp_fwd_c(ctx, x, ln_w, ln_b, w1, b1, w2, b2, eps, gelu, conv1d, hidden, drop)
        ctx.up = _upstream_of(x) if pre_ln else None
        mean, rstd = _stats(R if pre_ln else 0, dev)
        if pre_ln:
            xn = Act(R, d, dev)
            ops.layernorm_fwd(x, ln_w, ln_b, eps, xn.live, mean, rstd)
        else:  # bare MLP module (models/transformerblock.py:84-93)
            xn = to_act(x)
        u, h = Act(R, hidden, dev), Act(R, hidden, dev)
        _lin_fwd(xn, w1, conv1d, h.live, bias=b1, act=_GELU[gelu][0], pre=u.live)
        y = torch.empty(R, d_out, dtype=torch.float32, device=dev)
        _lin_fwd(h, w2, conv1d, y, bias=b2, residual=x if pre_ln else None, drop=_out_drop(drop))
        ctx.save_for_backward(x, ln_w, ln_b, w1, b1, w2, b2, mean, rstd)
        ctx.acts = (xn, u, h)
        ctx.cfg = (gelu, conv1d, hidden, pre_ln, drop)
        _note_output(y, _out_drop(drop), b2)
        return y

    @staticmethod
    @_in_backward_precision
    def backward(ctx, dy):
        if ctx.composite:
            return _mlp_bwd_c(ctx, dy)
        x, ln_w, ln_b, w1, b1, w2, b2, mean, rstd = ctx.saved_tensors
        xn, u, h = (_b16(t) for t in ctx.acts)
        gelu, conv1d, hidden, pre_ln, drop = ctx.cfg
        R, d = x.shape
        dev = x.device
        dy = dy.contiguous()
        od = _out_drop(drop)
        sh = _take_shadow(dy, od, b2)
        dya = sh.act if sh is not None else to_act(dy, od)
        with _Side(dev):
            g_w2 = _wgrad(dya, h, w2, conv1d)
            g_b2 = _accept_bias(sh) if sh is not None else _bgrad(dy if od is None else dya.live, b2)
        du = Act(R, hidden, dev)
        _lin_dgrad(dya, w2, conv1d, du.live, act=_GELU[gelu][1], aux=u.live)
        with _Side(dev):
            g_w1 = _wgrad(du, xn, w1, conv1d)
            g_b1 = _bgrad(du.live, b1)
        if pre_ln:
            dxn = Act(R, d, dev)
            _lin_dgrad(du, w1, conv1d, dxn.live)
            dx, g_lw, g_lb = _ln_bwd(dxn.live, x, ln_w, ln_b, mean, rstd, dx_in=dy, up=ctx.up)
        else:
            dx = torch.empty(R, d, dtype=torch.float32, device=dev)
            _lin_dgrad(du, w1, conv1d, dx)
            g_lw = g_lb = None
        join_side(dev)
        ctx.acts = None
        flush_ready()
        return dx, g_lw, g_lb, g_w1, g_b1, g_w2, g_b2, None, None, None, None, None


# --------------------------------------------------------------------------- pre-LN cross-attention sub-layer
class CrossAttnSublayer(torch.autograd.Function):
    """y = x + proj(attn(q = w_q LN_q(x), k = w_k LN_kv(mem), v = w_v LN_kv(mem)))
    -- DecoderBlock's cross-attention half, models/transformerblock.py:56-76,160."""

    @staticmethod
    def forward(ctx, x, mem, nq_w, nq_b, nkv_w, nkv_b, w_q, w_k, w_v, w_proj, b_proj, L, H, mask, eps, pre_ln=True,
                scale=None, drop=None, b_q=None, b_k=None, b_v=None):
        """b_q / b_k / b_v: CrossAttention(qkv_bias=True); mem may be narrower or wider than x (mem_dim != dim: w_k, w_v are [d, mem_dim]) --
        both off the AFFT configurations (models/transformerblock.py:41-50) and served call by call."""
        R, d = x.shape
        dm = mem.shape[1]
        nseq, hd = R // L, d // H
        dev = x.device
        ctx.composite = False
        ctx.qkv_bias = b_q is not None or b_k is not None or b_v is not None
        if (mask_table(mask) is None and not ctx.qkv_bias and dm == d and _composite_ok(x, pre_ln, d, f16x2=False)
                and mem.stride(0) == d):
            return _cross_fwd_c(ctx, x, mem, nq_w, nq_b, nkv_w, nkv_b, w_q, w_k, w_v, w_proj, b_proj, L, H, mask, eps, scale,
                                drop)
        ctx.up = _upstream_of(x) if pre_ln else None
        mq, rq = _stats(R if pre_ln else 0, dev)
        mk, rk = _stats(R if pre_ln else 0, dev)
        if pre_ln:
            xq, mkv = Act(R, d, dev), Act(R, dm, dev)
            ops.layernorm_fwd(x, nq_w, nq_b, eps, xq.live, mq, rq)
            ops.layernorm_fwd(mem, nkv_w, nkv_b, eps, mkv.live, mk, rk)
        else:  # bare CrossAttention module (models/transformerblock.py:56-76)
            xq, mkv = to_act(x), to_act(mem)
        q, k, v = Act(R, d, dev), Act(R, d, dev), Act(R, d, dev)
        _lin_fwd(xq, w_q, False, q.live, bias=b_q)
        _lin_fwd(mkv, w_k, False, k.live, bias=b_k)
        _lin_fwd(mkv, w_v, False, v.live, bias=b_v)
        ao = Act(R, d, dev)
        probs = torch.empty(nseq, H, L, L, dtype=torch.float32, device=dev)
        scale = float(scale) if scale else float(hd) ** -0.5
        _attention_fwd(q.live, k.live, v.live, nseq, L, H, hd, scale, mask, ao.live, probs, drop)
        y = torch.empty(R, d, dtype=torch.float32, device=dev)
        _lin_fwd(ao, w_proj, False, y, bias=b_proj, residual=x if pre_ln else None, drop=_out_drop(drop))
        ctx.save_for_backward(x, mem, nq_w, nq_b, nkv_w, nkv_b, w_q, w_k, w_v, w_proj, b_proj, mq, rq, mk, rk, probs)
        ctx.acts = (xq, mkv, q, k, v, ao)
        ctx.cfg = (L, H, scale, pre_ln, drop)
        ctx.qkv_b = (b_q, b_k, b_v)
        _note_output(y, _out_drop(drop), b_proj)
        return y

    @staticmethod
    @_in_backward_precision
    def backward(ctx, dy):
        if ctx.composite:
            return _cross_bwd_c(ctx, dy)
        (x, mem, nq_w, nq_b, nkv_w, nkv_b, w_q, w_k, w_v, w_proj, b_proj, mq, rq, mk, rk, probs) = ctx.saved_tensors
        xq, mkv, q, k, v, ao = (_b16(t) for t in ctx.acts)
        L, H, scale, pre_ln, drop = ctx.cfg
        R, d = x.shape
        nseq, hd = R // L, d // H
        dev = x.device
        dy = dy.contiguous()
        od = _out_drop(drop)
        sh = _take_shadow(dy, od, b_proj)
        dya = sh.act if sh is not None else to_act(dy, od)
        with _Side(dev):
            g_wp = _wgrad(dya, ao, w_proj, False)
            g_bp = _accept_bias(sh) if sh is not None else _bgrad(dy if od is None else dya.live, b_proj)
        dao = Act(R, d, dev)
        _lin_dgrad(dya, w_proj, False, dao.live)
        dq, dk, dv = Act(R, d, dev), Act(R, d, dev), Act(R, d, dev)
        ops.attention_bwd(dao.live, q.live, k.live, v.live, probs, nseq, L, H, hd, scale, dq.live, dk.live, dv.live,
                          *(_attn_drop(drop)))
        b_q, b_k, b_v = ctx.qkv_b
        with _Side(dev):
            g_q = _wgrad(dq, xq, w_q, False)
            g_k = _wgrad(dk, mkv, w_k, False)
            g_v = _wgrad(dv, mkv, w_v, False)
            g_bq = _bgrad(dq.live, b_q) if b_q is not None else None
            g_bk = _bgrad(dk.live, b_k) if b_k is not None else None
            g_bv = _bgrad(dv.live, b_v) if b_v is not None else None
        dmkv = torch.empty(R, mem.shape[1], dtype=torch.float32, device=dev)
        _lin_dgrad(dk, w_k, False, dmkv)
        _lin_dgrad(dv, w_v, False, dmkv, accumulate=True)
        if pre_ln:
            dxq = Act(R, d, dev)
            _lin_dgrad(dq, w_q, False, dxq.live)
            dmem, g_kw, g_kb = _ln_bwd(dmkv, mem, nkv_w, nkv_b, mk, rk, dx_in=None)
            dx, g_qw, g_qb = _ln_bwd(dxq.live, x, nq_w, nq_b, mq, rq, dx_in=dy, up=ctx.up)
        else:
            dx = torch.empty(R, d, dtype=torch.float32, device=dev)
            _lin_dgrad(dq, w_q, False, dx)
            dmem, g_qw, g_qb, g_kw, g_kb = dmkv, None, None, None, None
        join_side(dev)
        ctx.acts = None
        flush_ready()
        return dx, dmem, g_qw, g_qb, g_kw, g_kb, g_q, g_k, g_v, g_wp, g_bp, None, None, None, None, None, None, None, g_bq, g_bk, g_bv


# --------------------------------------------------------------------------- plain linear (mapping, enc/dec, classifier)
class Linear(torch.autograd.Function):
    """y[rows, out] = x @ W^T (+ b), fp32 in / fp32 out: feature mapping (models/feature_mapping.py:59-61),
    dim_encoder / dim_decoder (future_prediction.py:246-255) and the classifier (future_prediction.py:108-121)."""

    @staticmethod
    def forward(ctx, x, W, b, in_drop=None):
        rows = x.shape[0]
        n_out = W.shape[0]
        _forget_output()
        xa = to_act(x, in_drop)   # Dropout(p) on the input: classifier (future_prediction.py:108)
        ctx.in_drop = in_drop
        ybuf = torch.empty(rows, rt.pad64(n_out) if n_out % 4 else n_out, dtype=torch.float32, device=x.device)
        y = ybuf[:, :n_out]
        _lin_fwd(xa, W, False, y, bias=b)
        ctx.save_for_backward(W, b)
        ctx.xa = xa
        return y

    @staticmethod
    @_in_backward_precision
    def backward(ctx, dy):
        W, b = ctx.saved_tensors
        xa = _b16(ctx.xa)
        _drop_shadow()
        dya = to_act(dy)
        with _Side(dy.device):
            g_w = _wgrad(dya, xa, W, False)
            g_b = _bgrad(dya.live if rt.fp32_acts() else dy if dy.stride(1) == 1 else dy.contiguous(), b)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(xa.rows, W.shape[1], dtype=torch.float32, device=dy.device)
            _lin_dgrad(dya, W, False, dx)
            if ctx.in_drop is not None:
                ops.cast(dx, dx, drop=ctx.in_drop)   # replay the input mask on the gradient (in place)
        join_side(dy.device)
        ctx.xa = None
        flush_ready()
        return dx, g_w, g_b, None


# --------------------------------------------------------------------------- dense layer with a fused activation
_ACT_ID = {"none": ACT_NONE, "gelu": ACT_GELU_ERF, "relu": ACT_RELU, "gate": ACT_SIGMOID_GATE}


class LinearAct(torch.autograd.Function):
    """y = drop(act(x W^T + b)), act in none | relu | gelu (exact erf) | gate (y = aux * sigmoid(x W^T + b)), all in the
    GEMM epilogue: the MATT layers (models/fusion.py:40-46: Linear, ReLU, Dropout 0.8), the NonLinear mapping
    (models/feature_mapping.py:91-101) and ContextGating = glu(cat(x, fc(x))) of GatedLinear (:22-31, aux = x).
    Backward: one element-wise kernel turns dy into the GEMM operand d(pre) (and d(aux) for the gate)."""

    @staticmethod
    def forward(ctx, x, W, b, act, aux, out_drop):
        rows, n_out = x.shape[0], W.shape[0]
        dev = x.device
        _forget_output()
        xa = to_act(x)
        ybuf = torch.empty(rows, rt.pad64(n_out) if n_out % 4 else n_out, dtype=torch.float32, device=dev)
        y = ybuf[:, :n_out]
        pre = Act(rows, n_out, dev) if act in ("gelu", "gate") else None
        if aux is not None:
            aux = aux.contiguous()
        _lin_fwd(xa, W, False, y, bias=b, act=_ACT_ID[act], aux=aux, pre=None if pre is None else pre.live, drop=out_drop)
        ctx.save_for_backward(W, b, aux, y if act == "relu" else None)
        ctx.acts = (xa, pre)
        ctx.cfg = (act, out_drop)
        return y

    @staticmethod
    @_in_backward_precision
    def backward(ctx, dy):
        W, b, aux, y = ctx.saved_tensors
        xa, pre = ctx.acts
        xa = _b16(xa)
        act, out_drop = ctx.cfg
        rows, n_out = xa.rows, W.shape[0]
        dev = dy.device
        _drop_shadow()
        dy = dy if dy.stride(-1) == 1 else dy.contiguous()
        dpre = Act(rows, n_out, dev)
        daux = torch.empty(rows, n_out, dtype=torch.float32, device=dev) if act == "gate" else None
        saved = y if act == "relu" else (pre.live if pre is not None else None)
        ops.act_bwd(_ACT_ID[act], dy, saved, dpre.live, aux, daux, out_drop)
        with _Side(dev):
            g_w = _wgrad(dpre, xa, W, False)
            g_b = _bgrad(dpre.live, b)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(rows, W.shape[1], dtype=torch.float32, device=dev)
            _lin_dgrad(dpre, W, False, dx)
        join_side(dev)
        ctx.acts = None
        flush_ready()
        return dx, g_w, g_b, None, daux, None


class SoftmaxSmall(torch.autograd.Function):
    """softmax over the last (<= 32-wide) dimension of an fp32 [rows, n] matrix: MATT's modality weights
    (models/fusion.py:57)."""

    @staticmethod
    def forward(ctx, x):
        _forget_output()
        x = x if x.stride(-1) == 1 else x.contiguous()
        y = torch.empty(x.shape[0], x.shape[1], dtype=torch.float32, device=x.device)
        ops.softmax_small_fwd(x, y)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        _drop_shadow()
        dx = torch.empty_like(y)
        ops.softmax_small_bwd(y, dy if dy.stride(-1) == 1 else dy.contiguous(), dx)
        return dx


class WeightedSum(torch.autograd.Function):
    """out[r, :] = sum_i w[r, i] * xs[i][r, :]: the score fusion of CMFPScoreFusion (models/future_prediction.py:343-350);
    xs are the per-modality logits (fp32 [rows, classes], one common row stride), w the MATT weights [rows, M]."""

    @staticmethod
    def forward(ctx, w, *xs):
        _forget_output()
        w = w.contiguous()
        ld = xs[0].stride(0)
        xs = tuple(x if (x.stride(-1) == 1 and x.stride(0) == ld) else x.contiguous() for x in xs)
        if any(x.stride(0) != xs[0].stride(0) for x in xs):
            xs = tuple(x.contiguous() for x in xs)
        out = torch.empty(xs[0].shape[0], xs[0].shape[1], dtype=torch.float32, device=w.device)
        ops.weighted_sum_fwd(xs, w, out)
        ctx.save_for_backward(w, *xs)
        return out

    @staticmethod
    def backward(ctx, dout):
        w, *xs = ctx.saved_tensors
        _drop_shadow()
        dout = dout if dout.stride(-1) == 1 else dout.contiguous()
        dxs = [torch.empty(x.shape[0], x.shape[1], dtype=torch.float32, device=w.device) for x in xs]
        dw = torch.empty_like(w)
        ops.weighted_sum_bwd(xs, w, dout, dxs, dw)
        return (dw, *dxs)


# --------------------------------------------------------------------------- standalone LayerNorm (final norms)
class LayerNormRows(torch.autograd.Function):
    """y[r] = LN(X[r*stride_rows]) for r < rows: stride_rows = S picks token 0 of every frame
    (models/fusion.py:362-364), stride_rows = 1 is a plain LayerNorm (GPT-2 ln_f, CA-Fuser norm)."""

    @staticmethod
    def forward(ctx, X, w, b, eps, stride_rows):
        Rall, d = X.shape
        rows = Rall // stride_rows
        _forget_output()
        xv = X.view(rows, stride_rows * d)[:, :d]
        mean, rstd = _stats(rows, X.device)
        y = torch.empty(rows, d, dtype=torch.float32, device=X.device)
        ops.layernorm_fwd(xv, w, b, eps, y, mean, rstd)
        ctx.save_for_backward(X, w, b, mean, rstd)
        ctx.stride_rows = stride_rows
        return y

    @staticmethod
    def backward(ctx, dy):
        X, w, b, mean, rstd = ctx.saved_tensors
        _drop_shadow()
        s = ctx.stride_rows
        Rall, d = X.shape
        rows = Rall // s
        dy = dy.contiguous()
        if s == 1:
            dX, gw, gb = _ln_bwd(dy, X, w, b, mean, rstd, dx_in=None)
        else:
            dX = torch.zeros_like(X)
            _, gw, gb = _ln_bwd(dy, X.view(rows, s * d)[:, :d], w, b, mean, rstd, dx_in=None,
                                dx_out=dX.view(rows, s * d)[:, :d])
        flush_ready()
        return dX, gw, gb, None, None


class TakeRows(torch.autograd.Function):
    """y[r] = X[r * stride_rows] as dense rows: token 0 of every frame.  The SA-Fuser's last block needs its MLP half on
    these rows only -- the fuser returns token 0 (models/fusion.py:362-365), so the other S - 1 rows of the last block's
    MLP never reach an output and receive an exactly-zero gradient; the reference computes them and throws them away.
    Backward: zeros elsewhere (the rows that attention still sees as keys / values get their gradient through it)."""

    @staticmethod
    def forward(ctx, X, stride_rows):
        Rall, d = X.shape
        rows = Rall // stride_rows
        _forget_output()
        ctx.shape = (Rall, d, stride_rows)
        X3 = X.view(rows, stride_rows, d)
        if _frames_ok(X3):
            y = torch.empty(rows, d, dtype=X.dtype, device=X.device)
            ops.gather_frames(y.view(rows, 1, d), [(X3, 0, 1, 0)])
            return y
        return X3[:, 0].contiguous()

    @staticmethod
    def backward(ctx, dy):
        Rall, d, s = ctx.shape
        _drop_shadow()
        dX = torch.empty(Rall, d, dtype=dy.dtype, device=dy.device)
        dy3 = dy.view(Rall // s, 1, d) if dy.is_contiguous() else dy.reshape(Rall // s, 1, d)
        if _frames_ok(dy3):
            ops.gather_frames(dX.view(Rall // s, s, d), [(dy3, 0, 1, 0)])      # token-0 rows <- dy, zeros elsewhere: one launch
        else:
            dX.zero_()
            dX.view(Rall // s, s * d)[:, :d].copy_(dy)
        return dX, None


def _frames_ok(t: Tensor) -> bool:
    """a (clips, frames, C) fp32 tensor ops.gather_frames can walk in place"""
    return (t.dtype == torch.float32 and t.dim() == 3 and t.stride(2) == 1 and t.shape[2] % 4 == 0 and t.stride(0) % 4 == 0
            and t.stride(1) % 4 == 0 and t.data_ptr() % 16 == 0)


class SeenThenPredicted(torch.autograd.Function):
    """whole = [z_1, z_hat_2 .. z_hat_{T+k}] (models/future_prediction.py:161-170: the observed first frame, then the predictions), and
    its two ends as views: past_futures = whole[:, :T], future = whole[:, T:].  One launch forward; backward, the gradients of the
    three (whole: the merged classifier heads; past_futures: the feature-regression loss; future) are added and split into dz (frame 0,
    zeros elsewhere) and dz_hat by one launch each -- autograd's own graph for cat + two slices is a cat, two zero fills, two copies
    and an add."""

    @staticmethod
    def forward(ctx, z, zh, T):
        ctx.set_materialize_grads(False)
        B, Tz, C = zh.shape
        ctx.cfg = (tuple(z.shape), tuple(zh.shape), T)
        if _frames_ok(z) and _frames_ok(zh):
            whole = torch.empty(B, 1 + Tz, C, dtype=torch.float32, device=zh.device)
            ops.gather_frames(whole, [(z, 0, 1, 0), (zh, 1, 1 + Tz, -1)])
        else:
            whole = torch.cat([z[:, :1], zh], dim=1)
        return whole, whole[:, :T], whole[:, T:]

    @staticmethod
    def backward(ctx, g_whole, g_past, g_fut):
        zs, zhs, T = ctx.cfg
        B, Tz, C = zhs
        gs = [g for g in (g_whole, g_past, g_fut) if g is not None]
        if not gs:
            return None, None, None
        dev, dt = gs[0].device, gs[0].dtype
        if all(_frames_ok(g) for g in gs):
            dz = dzh = None
            if ctx.needs_input_grad[0]:
                dz = torch.empty(zs, dtype=dt, device=dev)
                ops.gather_frames(dz, [(g, 0, 1, 0) for g in (g_whole, g_past) if g is not None])
            if ctx.needs_input_grad[1]:
                dzh = torch.empty(zhs, dtype=dt, device=dev)
                srcs = []
                if g_whole is not None:
                    srcs.append((g_whole, 0, Tz, 1))
                if g_past is not None and T > 1:
                    srcs.append((g_past, 0, T - 1, 1))
                if g_fut is not None and Tz > T - 1:
                    srcs.append((g_fut, T - 1, Tz, 1 - T))
                ops.gather_frames(dzh, srcs)
            return dz, dzh, None
        tot = torch.zeros(B, 1 + Tz, C, dtype=dt, device=dev)
        if g_whole is not None:
            tot += g_whole
        if g_past is not None:
            tot[:, :T] += g_past
        if g_fut is not None:
            tot[:, T:] += g_fut
        dz = torch.zeros(zs, dtype=dt, device=dev)
        dz[:, :1] = tot[:, :1]
        return dz, tot[:, 1:].contiguous(), None


# --------------------------------------------------------------------------- SA-Fuser token assembly
class AssembleTokens(torch.autograd.Function):
    """X[(b,t), s, :] = [modal_token | feats...] (+ modality_embedding) -- models/fusion.py:338-352."""

    @staticmethod
    def forward(ctx, token, mod_embed, T, frame_level, *feats):
        BT, d = feats[0].shape
        S = len(feats) + 1
        _forget_output()
        X = torch.empty(BT * S, d, dtype=torch.float32, device=feats[0].device)
        tok2d = token.view(-1, d)
        ops.assemble_tokens(list(feats), tok2d, d if frame_level else 0,
                            None if mod_embed is None else mod_embed.view(S, d), BT, T, d, X.view(BT, S, d))
        ctx.save_for_backward(token, mod_embed)
        ctx.cfg = (T, frame_level, S, BT, d, [f.requires_grad for f in feats])
        return X

    @staticmethod
    def backward(ctx, dX):
        token, mod_embed = ctx.saved_tensors
        T, frame_level, S, BT, d, needs = ctx.cfg
        _drop_shadow()
        dX = dX.contiguous()
        dX3 = dX.view(BT, S * d)
        g_tok = g_emb = None
        sink = rt.grad_mode() == "sink"
        # modal token gradient: sum over frames of token-0 rows (per frame position if frame-level)
        if sink:
            gt, acc = rt.SINK.grad_buffer(token)
            if not acc:
                gt.zero_()
        else:
            gt = torch.zeros_like(token)
            g_tok = gt
        ops.reduce_rows_periodic(dX3[:, :d], T if frame_level else 1, gt.view(-1, d))
        if sink:
            _ready(token)
        if mod_embed is not None:
            if sink:
                ge, acc = rt.SINK.grad_buffer(mod_embed)
                if not acc:
                    ge.zero_()
            else:
                ge = torch.zeros_like(mod_embed)
                g_emb = ge
            ops.reduce_rows_periodic(dX, S, ge.view(S, d))
            if sink:
                _ready(mod_embed)
        gf = [dX3[:, (i + 1) * d:(i + 2) * d] if n else None for i, n in enumerate(needs)]
        flush_ready()
        return (g_tok, g_emb, None, None, *gf)


# --------------------------------------------------------------------------- periodic row table (wpe / position embeddings)
class AddRowTable(torch.autograd.Function):
    """y[r] = x[r] + table[offset + r % period]  -- GPT-2 wpe (HF modeling_gpt2.py:576-577) and the CA-Fuser
    position embedding (models/fusion.py:255-262)."""

    @staticmethod
    def forward(ctx, x, table, period, offset):
        _forget_output()
        y = torch.empty_like(x)
        ops.add_rows_periodic(x, table[offset:offset + period], period, y)
        ctx.save_for_backward(table)
        ctx.cfg = (period, offset)
        return y

    @staticmethod
    def backward(ctx, dy):
        (table,) = ctx.saved_tensors
        period, offset = ctx.cfg
        _drop_shadow()
        dy = dy.contiguous()
        if rt.grad_mode() == "sink" and table.is_leaf:   # a computed table (T-SA-Fuser) hands its gradient to autograd
            g, acc = rt.SINK.grad_buffer(table)
            if not acc:
                g.zero_()
            ops.reduce_rows_periodic(dy, period, g[offset:offset + period])
            _ready(table)
            flush_ready()
            return dy, None, None, None
        g = torch.zeros_like(table)
        ops.reduce_rows_periodic(dy, period, g[offset:offset + period])
        flush_ready()
        return dy, g, None, None


class SinkParam(torch.autograd.Function):
    """Identity on a (small) parameter that is consumed by torch glue ops (slicing / repeat of an embedding table):
    its gradient comes back through autograd and is routed into the gradient sink here, like every other parameter
    gradient of the path (first touch of a step overwrites, later ones add; readiness is notified), instead of being
    accumulated onto last step's value by autograd's AccumulateGrad."""

    @staticmethod
    def forward(ctx, p):
        ctx.save_for_backward(p)
        return p.view_as(p)

    @staticmethod
    def backward(ctx, g):
        (p,) = ctx.saved_tensors
        if rt.grad_mode() != "sink":
            return g
        buf, acc = rt.SINK.grad_buffer(p)
        if acc:
            buf.add_(g)
        else:
            buf.copy_(g)
        _ready(p)
        flush_ready()
        return None


class ScatterTokens(torch.autograd.Function):
    """X[g, s, :] = feats[s][g, :]: the modality features as the tokens of one sequence, without a modality token
    (models/fusion.py:107 CMFuser: (B*T, n, C); :176 T-SA-Fuser: (B, n*T, C) with rows = whole clips)."""

    @staticmethod
    def forward(ctx, *feats):
        G, W = feats[0].shape
        S = len(feats)
        _forget_output()
        X = torch.empty(G, S * W, dtype=torch.float32, device=feats[0].device)
        for i, f in enumerate(feats):
            ops.cast(f if f.stride(1) == 1 else f.contiguous(), X[:, i * W:(i + 1) * W])
        ctx.cfg = (S, W, [f.requires_grad for f in feats])
        return X

    @staticmethod
    def backward(ctx, dX):
        S, W, needs = ctx.cfg
        _drop_shadow()
        return tuple(dX[:, i * W:(i + 1) * W] if n else None for i, n in enumerate(needs))


class GroupMean(torch.autograd.Function):
    """y[g, :] = mean_s x[g, s, :] for x fp32 [G, S, W] (models/fusion.py:114, :207-210)."""

    @staticmethod
    def forward(ctx, x, G, S, W):
        _forget_output()
        ctx.in_shape = tuple(x.shape)
        x = x.contiguous()
        y = torch.empty(G, W, dtype=torch.float32, device=x.device)
        ops.group_sum(x, G, S, W, 1.0 / S, y)
        ctx.cfg = (G, S, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        G, S, W = ctx.cfg
        _drop_shadow()
        dx = torch.empty(G * S, W, dtype=torch.float32, device=dy.device)
        ops.group_bcast(dy.contiguous(), G, S, W, 1.0 / S, dx)
        return dx.view(ctx.in_shape), None, None, None


class ElementDropout(torch.autograd.Function):
    """nn.Dropout(p) on an fp32 [rows, d] tensor (token / embedding dropout, models/fusion.py:355,
    HF GPT2Model drop).  The mask is a pure function of (key, element index); backward replays it."""

    @staticmethod
    def forward(ctx, x, desc):
        _forget_output()
        y = torch.empty_like(x)
        ops.cast(x, y, drop=desc)
        ctx.desc = desc
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        ops.cast(dy, dx, drop=ctx.desc)
        flush_ready()
        return dx, None


# --------------------------------------------------------------------------- losses
class GradLanding:
    """Where the gradients of the two halves SplitRows hands out are to be written: ONE buffer of the whole tensor's shape, so that
    a loss kernel that knows the record (SoftmaxCE) stores its gradient in place and SplitRows.backward has nothing left to copy."""

    def __init__(self, shape, n):
        self.shape, self.n, self.dx = tuple(shape), n, None
        self.written = [False, False]      # per half: a loss kernel has stored its gradient there in THIS backward pass

    def half(self, which: int, device, dtype=torch.float32) -> Optional[Tensor]:
        """the half's slice of the shared buffer for ONE writer per backward pass; None for a second one (two losses on the same
        half: each must return its own tensor, autograd adds them -- a second in-place write would replace the first and the sum of
        the two aliasing views would come out as twice the last gradient)"""
        if self.written[which]:
            return None
        if self.dx is None:
            self.dx = torch.empty(self.shape, dtype=dtype, device=device)
        self.written[which] = True
        return self.dx[:, :self.n] if which == 0 else self.dx[:, self.n:]

    def holds(self, which: int, g: Tensor) -> bool:
        if self.dx is None or g is None:
            return False
        v = self.dx[:, :self.n] if which == 0 else self.dx[:, self.n:]
        return g.data_ptr() == v.data_ptr() and g.shape == v.shape and g.stride() == v.stride() and g.dtype == v.dtype


class SoftmaxCE(torch.autograd.Function):
    """Per-row softmax cross-entropy (reduction='none'); ignored rows give 0.  common/runner.py:13-37.
    logits: [rows, C], or a (clips, frames, C) view with contiguous classes (a half of the merged classifier output): walked in
    place.  landing = (GradLanding, half) makes backward write the gradient straight into the record's buffer."""

    @staticmethod
    def forward(ctx, logits, labels, soft, keep, landing=None):
        ctx.landing = landing
        if logits.dim() == 3:
            clips, frames, C = logits.shape
            assert logits.stride(2) == 1
            row_loss = torch.empty(clips * frames, dtype=torch.float32, device=logits.device)
            ops.softmax_ce_frames(logits, C, labels=labels, soft=soft, keep=keep, row_loss=row_loss)
            ctx.save_for_backward(logits, labels, soft, keep)
            return row_loss
        rows, C = logits.shape
        row_loss = torch.empty(rows, dtype=torch.float32, device=logits.device)
        lg = logits if logits.stride(1) == 1 else logits.contiguous()
        ops.softmax_ce(lg, C, labels=labels, soft=soft, keep=keep, row_loss=row_loss)
        ctx.save_for_backward(lg, labels, soft, keep)
        return row_loss

    @staticmethod
    def backward(ctx, g_rows):
        lg, labels, soft, keep = ctx.saved_tensors
        if lg.dim() == 3:
            C = lg.shape[2]
            d = None
            if ctx.landing is not None and ctx.landing[0].shape[2] == C:
                d = ctx.landing[0].half(ctx.landing[1], lg.device)
                assert d is None or d.shape == lg.shape
            if d is None:
                d = torch.empty(lg.shape, dtype=torch.float32, device=lg.device)
            ops.softmax_ce_frames(lg, C, labels=labels, soft=soft, keep=keep, dlogits3=d, row_g=g_rows.contiguous())
            flush_ready()
            return d, None, None, None, None
        rows, C = lg.shape
        d = torch.empty(rows, C, dtype=torch.float32, device=lg.device)
        ops.softmax_ce(lg, C, labels=labels, soft=soft, keep=keep, dlogits=d, row_g=g_rows.contiguous())
        flush_ready()
        return d, None, None, None, None


class MSE(torch.autograd.Function):
    """mean((a-b)^2), gradient to both sides (common/runner.py:164-166).
    a, b: 2-D [rows, d], or -- `a_lo` given -- 3-D (B, Ta, C) / (B, Tb, C) of which the frames a[:, a_lo : a_lo + nt] and
    b[:, b_lo : b_lo + nt] are compared (the reference slices `[:, 1:]` before the loss, runner.py:164: here the kernel walks
    the slices in place through its row strides -- B rows of nt * C values -- and the gradients come back as FULL tensors of the
    inputs' shapes, zeros outside the slices: no slice copies forward, no slice-backward fill + copy per side)."""

    @staticmethod
    def forward(ctx, a, b, a_lo=None, b_lo=0, nt=0):
        ctx.rng = None
        if a_lo is not None:
            assert a.dim() == 3 and b.dim() == 3 and a.shape[0] == b.shape[0] and a.shape[2] == b.shape[2]
            assert a.stride(2) == 1 and b.stride(2) == 1 and a.stride(1) == a.shape[2] and b.stride(1) == b.shape[2]
            B, _, C = a.shape
            ctx.rng = (a_lo, b_lo, nt)
            av = torch.as_strided(a, (B, nt * C), (a.stride(0), 1), a.storage_offset() + a_lo * C)
            bv = torch.as_strided(b, (B, nt * C), (b.stride(0), 1), b.storage_offset() + b_lo * C)
        else:
            av, bv = a, b
        rows, d = av.shape
        out = torch.empty((), dtype=torch.float32, device=a.device)
        ops.mse_loss(av, bv, 1.0 / (rows * d), out)
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        if ctx.rng is None:
            rows, d = a.shape
            da = torch.zeros(rows, d, dtype=torch.float32, device=a.device)
            db = torch.zeros(rows, d, dtype=torch.float32, device=a.device)
            ops.mse(a, b, 1.0 / (rows * d), None, da, db, g_dev=g.contiguous())
            flush_ready()
            return da, db
        a_lo, b_lo, nt = ctx.rng
        B, _, C = a.shape
        need_a, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        da = torch.empty(a.shape, dtype=torch.float32, device=a.device) if need_a else None
        db = torch.empty(b.shape, dtype=torch.float32, device=a.device) if need_b else None
        if C % 4 == 0 and a.stride(0) % 4 == 0 and b.stride(0) % 4 == 0:
            ops.mse_frames_bwd(a, b, a_lo, b_lo, nt, 1.0 / (B * nt * C), g.contiguous(), da, db)      # writes zeros outside the ranges
        else:
            av = torch.as_strided(a, (B, nt * C), (a.stride(0), 1), a.storage_offset() + a_lo * C)
            bv = torch.as_strided(b, (B, nt * C), (b.stride(0), 1), b.storage_offset() + b_lo * C)
            dav = dbv = None
            if da is not None:
                dav = torch.as_strided(da.zero_(), (B, nt * C), (da.stride(0), 1), a_lo * C)
            if db is not None:
                dbv = torch.as_strided(db.zero_(), (B, nt * C), (db.stride(0), 1), b_lo * C)
            ops.mse(av, bv, 1.0 / (B * nt * C), None, dav, dbv, g_dev=g.contiguous())
        flush_ready()
        return da, db, None, None, None


class ReduceLosses(torch.autograd.Function):
    """Runner._reduce_loss (common/runner.py:198-213) in one launch each way: total = sum_i w_i * mean(v_i) over the per-row
    loss tensors v_i (scalars count as one row), and the means for the log.  Backward hands every producer its upstream
    gradient g * w_i / n_i as one filled buffer (the softmax-CE kernel's row_g, the MSE kernel's g_dev) -- instead of
    mean / mul / stack / sum forward and div / mul / expand / copy backward per term (15 torch kernels per step at three terms)."""

    @staticmethod
    def forward(ctx, weights, *vals):
        flat = [v.reshape(-1).float().contiguous() for v in vals]
        dev = flat[0].device
        means = torch.empty(len(flat), dtype=torch.float32, device=dev)
        total = torch.empty((), dtype=torch.float32, device=dev)
        ops.loss_reduce(flat, weights, means, total)
        ctx.weights, ctx.shapes = tuple(weights), [v.shape for v in vals]
        ctx.total = total.detach()      # backward's kernel tells the optimizer kernels whether the step is finite (runtime.SINK.step_ok)
        ctx.needs = list(ctx.needs_input_grad[1:])
        ctx.mark_non_differentiable(means)
        ctx.set_materialize_grads(False)      # `means` takes no gradient: without this autograd zero-fills one for it every step
        return total, means

    @staticmethod
    def backward(ctx, g_total, _g_means):
        if g_total is None:
            return (None,) * (1 + len(ctx.shapes))
        grads = [torch.empty(shp, dtype=torch.float32, device=g_total.device) if need else None
                 for shp, need in zip(ctx.shapes, ctx.needs)]
        ops.loss_reduce_bwd(grads, ctx.weights, g_total.contiguous(), total=ctx.total, ok=rt.SINK.step_ok)
        return (None, *grads)


class SplitRows(torch.autograd.Function):
    """(x[:, :n], x[:, n:]) of x (B, L, C) as two views whose gradients are written side by side into ONE buffer: autograd's own
    slicing would zero-fill a full-size gradient per slice, copy into it and add the two (5 kernels on the (B, T + 1, 3806) logits
    of the merged classifier heads).  With a GradLanding record (split_rows below) the loss kernels write their gradients into that
    buffer themselves and backward launches nothing; a gradient that arrives from anywhere else is copied in."""

    @staticmethod
    def forward(ctx, x, n, landing=None):
        ctx.n, ctx.shape, ctx.landing = n, x.shape, landing
        return x[:, :n], x[:, n:]

    @staticmethod
    def backward(ctx, ga, gb):
        ref = ga if ga is not None else gb
        rec = ctx.landing
        if rec is not None and rec.dx is not None and rec.dx.dtype == ref.dtype:
            dx = rec.dx
        else:
            dx, rec = torch.empty(ctx.shape, dtype=ref.dtype, device=ref.device), None
        for which, g, view in ((0, ga, dx[:, :ctx.n]), (1, gb, dx[:, ctx.n:])):
            if g is None:
                view.zero_()
            elif rec is None or not rec.holds(which, g):
                view.copy_(g)
        if ctx.landing is not None:
            ctx.landing.written = [False, False]      # the next backward pass through this record starts clean
        return dx, None, None


def split_rows(x: Tensor, n: int):
    """SplitRows with the landing record attached to both halves (`_afft_landing`: read by MultiDimCrossEntropy)"""
    rec = GradLanding(x.shape, n)
    a, b = SplitRows.apply(x, n, rec)
    a._afft_landing, b._afft_landing = (rec, 0), (rec, 1)
    return a, b

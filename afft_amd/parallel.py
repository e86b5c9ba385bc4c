"""Data-parallel training runtime for the hot path: one process per GPU (torchrun), full weight replica per
GPU, one exchange per step -- the gradient sum -- carried by RCCL over xGMI (torch.distributed backend
"nccl" IS RCCL on ROCm), overlapped with the remaining backward.

Replaces the reference's torch DDP wrapper + per-parameter SGD groups (train.py:189-225,352,364-368,251-265):

  * parameters, gradients and momentum live in three flat fp32 buffers (views handed back to the modules), so
    the optimizer is a fused Nesterov-SGD kernel over contiguous slices and a bucket is a contiguous slice;
  * weight-gradient GEMMs write straight into the flat gradient buffer (afft_amd.runtime.GradSink); when the last
    gradient of a bucket has been produced the bucket is handed to a SIDE STREAM: (world > 1) its all-reduce --
    xGMI is point-to-point (7 links/GPU), so buckets are large (>= 128 MB) and few, and the payload can be sent
    as bf16 (``comm_dtype``) to halve the per-link bytes -- followed immediately by the SGD update of exactly that
    slice (HBM-bound, it overlaps the MFMA-bound backward GEMMs of the earlier layers) which also rewrites the bf16
    weight images of the slice;
  * nothing in the data path needs a collective besides that sum (clips are independent).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import os
import torch
import torch.distributed as dist

from . import ops, runtime as rt

Tensor = torch.Tensor


def _align(n: int, a: int = 64) -> int:
    return (n + a - 1) // a * a


def mark_fp32_tables(module: torch.nn.Module):
    """Tag every parameter the kernels read as fp32 rows, never as a GEMM operand image: the weights of nn.Embedding modules
    (models/future_prediction.py `wpe`, models/fusion.py `position_embeddings`: functional.AddRowTable adds their fp32 rows).  The
    mirrored modules tag their own tables at construction (so a bare parameter list, afft_amd.optim.SGD, is classified the same
    way); this walk covers any other nn.Embedding of a foreign module."""
    for m in module.modules():
        if isinstance(m, torch.nn.Embedding):
            m.weight._afft_fp32_table = True


class FlatParams:
    """Re-homes every trainable parameter of `model` (and its .grad) in contiguous fp32 buffers."""

    def __init__(self, model, sharded_layout: bool = False):
        """model: an nn.Module (its trainable parameters, in module order) or an iterable of parameters (the optimizer's
        param_groups, flattened: afft_amd.optim.SGD).
        sharded_layout (the N > 1 sharded update, GradReducer comm_algo = 'sharded'): the GEMM weights that own a bf16 image in
        the flat buffers come FIRST (module order), everything else -- biases, LayerNorm weights, tokens, odd-shaped matrices:
        what the kernels read as fp32 every step -- behind them, from element `split` on.  The first region is updated in 1 / N
        slices (only its 16-bit images have to be whole on every rank), the second stays replicated."""
        if isinstance(model, torch.nn.Module):
            mark_fp32_tables(model)
        src = model.parameters() if isinstance(model, torch.nn.Module) else model
        self.params: List[Tensor] = []
        seen = set()
        for p in src:
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                self.params.append(p)
        assert self.params, "no trainable parameters"
        self.sharded_layout = bool(sharded_layout)
        if self.sharded_layout:
            big = [p for p in self.params if self.owns_image(p)]
            self.params = big + [p for p in self.params if not self.owns_image(p)]
            self.n_big = len(big)
        else:
            self.n_big = 0
        dev = self.params[0].device
        self.offsets: List[int] = []
        off = 0
        for p in self.params:
            assert p.dtype == torch.float32 and p.device == dev
            self.offsets.append(off)
            off += _align(p.numel())
        self.total = off
        self.split = self.offsets[self.n_big] if self.n_big < len(self.params) else off      # first element of the replicated region
        if not self.sharded_layout:
            self.split = 0
        self.flat_p = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(off, dtype=torch.float32, device=dev)
        # bf16 image of every parameter at the same element offsets: written by the fused SGD kernel, so the
        # MFMA operand copies of the weights cost no extra pass (only GEMM weights with both dims % 64 == 0 use it;
        # odd shapes -- the 3806-row classifier, the 352-column objects mapping -- keep a padded cast image)
        self.flat_p16 = torch.zeros(off, dtype=torch.bfloat16, device=dev) if dev.type == "cuda" else None
        # fragment-packed bf16 images (same offsets and sizes; runtime.packed_images): the B operand of the "B direct" GEMM
        # kernels.  A slot per GEMM weight; an image goes live when a forward GEMM first wants it (runtime.weight_packed) and is
        # from then on written by the fused optimizer epilogue or re-packed after every other update (refresh_packed)
        # (the sharded update cannot keep them: a rank's slice of the flat buffer is not a slice of a packed image)
        use_pk = dev.type == "cuda" and rt.packed_images() and not self.sharded_layout
        self.flat_pk16 = torch.zeros(off, dtype=torch.bfloat16, device=dev) if use_pk else None
        self.packed: List[tuple] = []         # (offset, param, packed flat view)
        # FP16 images (same offsets; the B operands of the fp16 two-pass forward GEMMs, precision 'fp16x2'): allocated the first
        # time a step runs in that precision (ensure_f16) and from then on written by every optimizer kernel beside the bf16 ones
        self.flat_h16: Optional[Tensor] = None
        self.flat_p8: Optional[Tensor] = None      # e4m3(2^8 p) byte images (the fp8 lo pass of 'fp16x2': runtime.lo8), allocated with flat_h16
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                n = p.numel()
                self.flat_p[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[o:o + n].view(p.shape)
                p.grad = self.flat_g[o:o + n].view(p.shape)
            if self.flat_p16 is not None:
                ops.cast(self.flat_p.view(off // 64, 64), self.flat_p16.view(off // 64, 64))
                for p, o in zip(self.params, self.offsets):
                    if self.owns_image(p):
                        pk = None
                        if use_pk:
                            pk = self.flat_pk16[o:o + p.numel()]
                            self.packed.append((o, p, pk))
                        rt.adopt_weight_image(p, self.flat_p16[o:o + p.numel()].view(p.shape), packed=pk)
        rt.invalidate_weight_images()
        self.ensure_f16()

    @staticmethod
    def owns_image(p: Tensor) -> bool:
        """a GEMM weight whose 16-bit images live in the flat buffers (both dimensions multiples of 64; odd shapes -- the 3806-row
        classifier, the 352-column objects mapping -- keep a padded cast image that is redone from the fp32 master every step).
        A parameter the forward pass reads as fp32 -- an embedding table (GPT-2 `wpe` is [1024, 2048]), a token -- is never one
        whatever its shape (mark_fp32_tables): under the sharded update only the 16-bit images of the sharded region are whole on
        every rank, so everything read in fp32 must stay in the replicated region."""
        return (p.dim() == 2 and p.shape[0] % 64 == 0 and p.shape[1] % 64 == 0 and not getattr(p, "_afft_fp32_table", False))

    def ensure_f16(self):
        """the FP16 images exist once the precision is 'fp16x2' (called at construction and when a step begins)"""
        if self.flat_h16 is not None or rt.precision() != "fp16x2" or not self.flat_p.is_cuda:
            return
        n = self.total
        self.flat_h16 = torch.zeros(n, dtype=torch.float16, device=self.flat_p.device)
        with torch.no_grad():
            ops.cast(self.flat_p.view(n // 64, 64), self.flat_h16.view(n // 64, 64))
        if rt.lo8():
            self.flat_p8 = torch.zeros(n, dtype=torch.uint8, device=self.flat_p.device)
            with torch.no_grad():
                ops.quant_e4m3(self.flat_p.view(n // 64, 64), 256.0, self.flat_p8.view(n // 64, 64))
        for p, o in zip(self.params, self.offsets):
            img = getattr(p, "_afft_img", None)
            if img is not None and img.external and p.dim() == 2:
                rt.adopt_weight_f16(p, self.flat_h16[o:o + p.numel()].view(p.shape))
                if self.flat_p8 is not None:
                    rt.adopt_weight_f8(p, self.flat_p8[o:o + p.numel()].view(p.shape))

    def h16(self, s: int = 0, e: Optional[int] = None) -> Optional[Tensor]:
        return None if self.flat_h16 is None else self.flat_h16[s:self.total if e is None else e]

    def p8(self, s: int = 0, e: Optional[int] = None) -> Optional[Tensor]:
        return None if self.flat_p8 is None else self.flat_p8[s:self.total if e is None else e]

    def refresh_images(self):
        """Re-derive every bf16 image from the fp32 masters (after the masters were written from outside: a parameter
        broadcast, a checkpoint load into the flat buffer)."""
        if self.flat_p16 is not None:
            n = self.total
            with torch.no_grad():
                ops.cast(self.flat_p.view(n // 64, 64), self.flat_p16.view(n // 64, 64))
                if self.flat_h16 is not None:
                    ops.cast(self.flat_p.view(n // 64, 64), self.flat_h16.view(n // 64, 64))
                if self.flat_p8 is not None:
                    ops.quant_e4m3(self.flat_p.view(n // 64, 64), 256.0, self.flat_p8.view(n // 64, 64))
            self.refresh_packed(0, n)
        rt.invalidate_weight_images()

    def refresh_packed(self, s: int, e: int, skip=()):
        """Re-pack the fragment-packed images of the weights in flat range [s, e) from their fp32 masters (after an update that
        did not run in a weight-gradient epilogue); skip: ids of the weights whose epilogue keeps their image fresh itself."""
        if not self.packed:
            return
        with torch.no_grad():
            for o, p, pk in self.packed:
                if s <= o < e and id(p) not in skip and rt.packed_live(p):
                    ops.pack_weight(p.detach(), pk)

    def index_of(self) -> Dict[int, int]:
        return {id(p): i for i, p in enumerate(self.params)}


class GradReducer:
    """Bucketed, backward-overlapped gradient all-reduce over the flat gradient buffer.  `on_bucket` (optional) is
    called on the side stream right after a bucket's gradient is final (reduced): the per-bucket optimizer hook."""

    def __init__(self, flat: FlatParams, group=None, bucket_elems: int = 32 * 1024 * 1024, comm_dtype: str = "fp32",
                 force_comm: bool = False, comm_algo: str = "allreduce"):
        """comm_algo: 'allreduce' (one all-reduce per bucket), 'rs_ag' (reduce-scatter + all-gather of the same bucket: the
        same bytes per link as a ring all-reduce, in two collectives the library schedules independently -- the fallback
        for a fabric on which the all-reduce picks a slow algorithm; xGMI is point-to-point) or 'sharded'
        (the flat buffers must have the sharded layout): per bucket of GEMM weights the gradient is reduce-SCATTERED, every rank
        runs the update on its 1 / N slice of parameters, momentum and 16-bit images only, and the slices of the IMAGES -- what
        the next forward pass reads -- are all-gathered: 2 B instead of 4 B per parameter in the second half of the exchange and
        1 / N of the optimizer's HBM traffic.  The fp32 masters and momentum of the other ranks' slices go stale until
        sync_masters() (checkpoints, state_dict()).  The small replicated region behind FlatParams.split is all-reduced and
        updated everywhere as before.  Needs the update inside the backward pass (on_bucket); without it a step falls back to
        reduce-scatter + all-gather of the gradient."""
        if comm_algo not in ("allreduce", "rs_ag", "sharded"):
            raise ValueError("comm_algo must be 'allreduce', 'rs_ag' or 'sharded'")
        if comm_algo == "sharded" and not flat.sharded_layout:
            raise ValueError("comm_algo 'sharded' needs FlatParams(..., sharded_layout=True)")
        self.comm_algo = comm_algo
        self.flat = flat
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        # force_comm: run the collective path even in a 1-rank group (exercises RCCL + the bf16 payload on one GPU)
        self.comm = self.world > 1 or (force_comm and dist.is_available() and dist.is_initialized())
        self.comm_dtype = comm_dtype
        self.buckets: List[tuple] = []        # (start, end) element ranges, in parameter order
        self.bucket_of: List[int] = []        # param index -> bucket index
        start, cur = 0, 0
        split = flat.split if comm_algo == "sharded" else 0      # a bucket never straddles the sharded / replicated boundary
        for i, (p, o) in enumerate(zip(flat.params, flat.offsets)):
            if split and o == split and cur > start:
                self.buckets.append((start, cur))
                start = cur
            self.bucket_of.append(len(self.buckets))
            cur = o + _align(p.numel())
            if cur - start >= bucket_elems:
                self.buckets.append((start, cur))
                start = cur
        if cur > start:
            self.buckets.append((start, cur))
        self.masters_stale = False            # sharded: fp32 masters / momentum of the other ranks' slices are behind (sync_masters)
        self.opt_buf: Optional[Tensor] = None  # sharded: the optimizer's momentum buffer (set by whoever owns the optimizer), for sync_masters
        self.bucket_of = [min(b, len(self.buckets) - 1) for b in self.bucket_of]
        self._pidx = flat.index_of()
        self.expected: Optional[List[int]] = None     # ready-callbacks per bucket per step (learned on step 1)
        self._count = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._handles: List = []
        self._use_cuda = flat.flat_g.is_cuda
        self.step_ok: Optional[Tensor] = None     # the optimizer's "step is finite" flag (FusedSGD.ok), set by its owner: see _agree_ok
        self._ok_agreed = False
        dev_ = flat.flat_g.device
        if self._use_cuda:
            rt.aux_stream(dev_)      # exists before the streams below are drawn: they must differ from it (runtime.new_stream)
        self.side_stream = rt.new_stream(dev_) if self._use_cuda else None
        # with a gradient exchange the per-bucket optimizer gets a stream of its own: on one stream bucket b+1's all-reduce
        # would queue behind bucket b's update and the two (xGMI-bound and HBM-bound) would never overlap each other
        self.opt_stream = rt.new_stream(dev_, self.side_stream) if (self._use_cuda and self.comm) else None
        self.flat_g16 = (torch.empty(flat.total, dtype=torch.bfloat16, device=flat.flat_g.device)
                         if (comm_dtype == "bf16" and self.comm) else None)
        self.on_bucket: Optional[Callable[[int, int, Tensor, float], None]] = None
        self.defer_all = False     # True: nothing is handed over during backward, finish_step launches every bucket (callers whose
                                   # gradients only reach the flat buffer at the end of backward: DistributedDataParallel below)

    # ---- step protocol
    def begin_step(self):
        from . import functional as F_
        F_.settle_joins(self.flat.flat_g.device, drop_pending=True)      # a previous backward pass that raised left its end-of-pass join undone
        self.flat.ensure_f16()
        self._count = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._handles = []
        self._ok_agreed = False
        rt.SINK.begin_step()
        early = self.comm or self.on_bucket is not None
        rt.SINK.on_grad_ready = self._on_ready if early else None

    def _on_ready(self, p: Tensor):
        i = self._pidx.get(id(p))
        if i is None:
            return
        b = self.bucket_of[i]
        self._count[b] += 1
        if self.expected is not None and not self.defer_all and not self._launched[b] and self._count[b] == self.expected[b]:
            self._launch(b)

    def _grad_slice(self, s: int, e: int):
        if self.flat_g16 is not None:
            return self.flat_g16[s:e]
        return self.flat.flat_g[s:e]

    # ---- sharded update
    def sharded_bucket(self, b: int) -> bool:
        s, e = self.buckets[b]
        return (self.comm_algo == "sharded" and self.comm and self.on_bucket is not None and e <= self.flat.split
                and (e - s) % self.world == 0)

    def shard_of(self, s: int, e: int):
        """this rank's slice of bucket [s, e)"""
        n = (e - s) // self.world
        r = dist.get_rank(self.group)
        return s + r * n, s + (r + 1) * n

    def _launch_sharded(self, s: int, e: int):
        """reduce-scatter the bucket's gradient, update this rank's slice, all-gather the slices of the 16-bit images"""
        ss, se = self.shard_of(s, e)
        g = self._grad_slice(s, e)
        dist.reduce_scatter_tensor(g[ss - s:se - s], g, group=self.group)
        self.on_bucket(ss, se, g[ss - s:se - s], 1.0 / self.world)
        for img in (self.flat.flat_p16, self.flat.flat_h16, self.flat.flat_p8):
            if img is not None:
                dist.all_gather_into_tensor(img[s:e], img[ss:se], group=self.group)
        if self.flat.flat_p16 is None or rt.precision() not in ("bf16", "fp16x2"):
            # CPU tensors (gloo tests: no images), and the 'fp32' / 'bf16x3' precisions, whose GEMMs read the fp32 master (or its
            # split) of every weight (functional._lin_fwd): the fp32 slices themselves are what the next forward reads
            dist.all_gather_into_tensor(self.flat.flat_p[s:e], self.flat.flat_p[ss:se], group=self.group)
        self.masters_stale = True           # (the momentum of the other ranks' slices at least)

    def sync_masters(self):
        """Sharded update: bring the fp32 masters and the momentum of the other ranks' slices up to date (all-gather of every
        sharded bucket) -- before a checkpoint / state_dict(), or before anything else reads the fp32 values of a GEMM weight.
        A collective: every rank calls it."""
        if not self.masters_stale:
            return
        if self._use_cuda:
            torch.cuda.current_stream().wait_stream(self.side_stream)
            if self.opt_stream is not None:
                torch.cuda.current_stream().wait_stream(self.opt_stream)
        with torch.no_grad():
            for (s, e) in self.buckets:
                if e <= self.flat.split and (e - s) % self.world == 0:
                    ss, se = self.shard_of(s, e)
                    dist.all_gather_into_tensor(self.flat.flat_p[s:e], self.flat.flat_p[ss:se], group=self.group)
                    if self.opt_buf is not None:
                        dist.all_gather_into_tensor(self.opt_buf[s:e], self.opt_buf[ss:se], group=self.group)
        self.masters_stale = False

    def assert_masters_fresh(self, what: str):
        """state_dict() of a model / optimizer / DDP wrapper under the sharded update: the reference saves on the main process only
        (train.py:403-411 inside `if utils.is_main_process()`), so a collective here would be entered by rank 0 alone and hang or
        pair up with another rank's next reduce-scatter.  The masters are brought up to date where EVERY rank passes -- the first
        evaluation / no-grad forward after training steps (the reference validates on all ranks before it saves), or an explicit
        sync_masters() on every rank -- and a state_dict() that still finds them stale says so instead of communicating."""
        if self.masters_stale and self.world > 1:
            raise RuntimeError(f"{what}: the fp32 masters / momentum of the other ranks' slices are stale (comm_algo='sharded').  Call "
                               "sync_masters() on EVERY rank first (a collective), or run an evaluation forward on every rank "
                               "(model.eval() / torch.no_grad()), which does it; state_dict() itself never communicates")
        if self.masters_stale:
            self.sync_masters()

    def _agree_ok(self):
        """N > 1: the ranks agree on "this step is finite" (FusedSGD.ok) before the first update of the step -- a MIN over the
        flags every rank's first backward kernel wrote; the summed gradient of a step in which ANY rank saw a non-finite loss is
        non-finite on every rank, so all of them must skip.  One 4-byte collective per step, on the stream the updates follow."""
        if self.comm and self.step_ok is not None and not self._ok_agreed:
            dist.all_reduce(self.step_ok, op=dist.ReduceOp.MIN, group=self.group)
            self._ok_agreed = True

    def _launch(self, b: int):
        self._launched[b] = True
        s, e = self.buckets[b]
        if not self._use_cuda:     # CPU tensors (gloo; the build container's tests): same protocol, no streams
            self._agree_ok()
            if self.sharded_bucket(b):
                if self.flat_g16 is not None:
                    self.flat_g16[s:e].copy_(self.flat.flat_g[s:e])
                self._launch_sharded(s, e)
                return
            if self.comm:
                if self.flat_g16 is not None:      # bf16 payload
                    self.flat_g16[s:e].copy_(self.flat.flat_g[s:e])
                if self.comm_algo != "allreduce" or self.flat_g16 is not None or self.on_bucket is not None:
                    self._reduce(self._grad_slice(s, e))
                else:
                    self._handles.append(dist.all_reduce(self.flat.flat_g[s:e], group=self.group, async_op=True))
            if self.on_bucket is not None:
                self.on_bucket(s, e, self._grad_slice(s, e), 1.0 / self.world)
            return
        ev = torch.cuda.Event()
        ev.record()
        self.side_stream.wait_event(ev)
        if rt.overlap_wgrad():    # weight gradients of this bucket were enqueued on the auxiliary stream
            self.side_stream.wait_stream(rt.aux_stream(self.flat.flat_g.device))
        with torch.cuda.stream(self.side_stream):
            self._agree_ok()
            if self.comm and self.flat_g16 is not None:
                n = e - s
                ops.cast(self.flat.flat_g[s:e].view(n // 64, 64), self.flat_g16[s:e].view(n // 64, 64))
            if self.sharded_bucket(b):
                # one stream for the three steps of a bucket (reduce-scatter -> update of the slice -> all-gather of the images):
                # they depend on each other; bucket b + 1's collectives queue behind on the same stream, as RCCL would order them anyway
                self._launch_sharded(s, e)
                return
            if self.comm:
                # RCCL enqueues behind the work already on this stream; later work on this stream follows it
                self._reduce(self._grad_slice(s, e))
            if self.on_bucket is not None:
                if self.comm and self.opt_stream is not None:
                    done = torch.cuda.Event()
                    done.record()
                    self.opt_stream.wait_event(done)
                    with torch.cuda.stream(self.opt_stream):
                        self.on_bucket(s, e, self._grad_slice(s, e), 1.0 / self.world)
                else:
                    self.on_bucket(s, e, self._grad_slice(s, e), 1.0 / self.world)

    def _reduce(self, g: Tensor):
        """sum `g` (a contiguous bucket of the flat gradient buffer) over the ranks, in place"""
        if self.comm_algo == "allreduce" or g.numel() % self.world != 0 or (self.comm_algo == "sharded" and self.on_bucket is not None):
            dist.all_reduce(g, group=self.group)
            return
        n = g.numel() // self.world
        r = dist.get_rank(self.group)
        shard = g[r * n:(r + 1) * n]        # reduce-scatter into this rank's own slice of the bucket, all-gather in place
        dist.reduce_scatter_tensor(shard, g, group=self.group)
        dist.all_gather_into_tensor(g, shard, group=self.group)

    def finish_step(self):
        """Call after backward: zero untouched grads, hand over whatever is still pending, join the side stream."""
        from . import functional as F_
        F_.flush_ready(all_threads=True)       # a notification whose Function did not flush must not leak into the next step's counts
        F_.settle_joins(self.flat.flat_g.device)
        rt.SINK.on_grad_ready = None
        rt.SINK.finish_step(self.flat.params)
        if not self.comm and self.on_bucket is None:
            return
        if self.expected is None:
            self.expected = list(self._count)
            self._check_expected_agree()
        # fixed order on every rank: buckets are handed over last-to-first (the order backward completes them)
        for b in reversed(range(len(self.buckets))):
            if not self._launched[b]:
                self._launch(b)
        for h in self._handles:
            h.wait()
        if self._use_cuda:
            torch.cuda.current_stream().wait_stream(self.side_stream)
            if self.opt_stream is not None:
                torch.cuda.current_stream().wait_stream(self.opt_stream)

    def _check_expected_agree(self):
        """The readiness counts learned on step 1 decide WHEN a bucket's collective is enqueued during backward; the ORDER of
        the collectives must be the same on every rank or RCCL deadlocks.  With equal counts the order is a function of the
        backward graph alone (the same on every replica), so: once, when the counts are learned, every rank checks that it
        holds the same counts as the others (max == min == own) and raises instead of hanging later."""
        if not (self.comm and self.world > 1):
            return
        mine = torch.tensor(self.expected, dtype=torch.int64, device=self.flat.flat_g.device)
        hi, lo = mine.clone(), mine.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        if not (torch.equal(hi, mine) and torch.equal(lo, mine)):
            raise RuntimeError(f"GradReducer: ranks disagree on the per-bucket gradient counts of a step (here {self.expected}, "
                               f"max over ranks {hi.tolist()}, min {lo.tolist()}): the replicas do not run the same graph")

    def grad_for_optimizer(self):
        """(flat gradient tensor, scale): summed over ranks; the optimizer applies 1/world."""
        if self.flat_g16 is not None:
            return self.flat_g16, 1.0 / self.world
        return self.flat.flat_g, 1.0 / self.world


class FusedSGD:
    """Momentum SGD over the flat buffers (conf/opt/optimizer/sgd.yaml + expts/01: lr 1e-3, momentum 0.9, nesterov, wd 1e-6):
    one kernel launch per contiguous slice instead of 151 parameter groups; the same kernel writes the bf16 weight images of
    the slice.  `hyper` (optional, one (lr, weight_decay) per parameter of `flat`, set_hyper) carries the per-module values
    train.py:189-225 allows: parameters are then updated class by class, one launch per distinct (lr, wd) and bucket."""

    def __init__(self, flat: FlatParams, lr: float = 1e-3, momentum: float = 0.9, weight_decay: float = 1e-6,
                 nesterov: bool = True):
        self.flat, self.lr, self.momentum, self.wd, self.nesterov = flat, lr, momentum, weight_decay, nesterov
        self.buf = torch.zeros_like(flat.flat_p)
        # "the step is finite" (device float; None = no guard): the loss reduction of a step writes isfinite(loss) here when the
        # owner has put it into runtime.SINK.step_ok, and every update kernel of that step -- bucket kernels and weight-gradient
        # epilogues alike -- leaves parameters, momentum and images untouched when it reads 0.  The reference raises 'The loss is
        # NaN!' before backward (common/runner.py:209); with lazy metrics that error surfaces a step later, and this keeps the
        # state it finds as the reference would have left it.  N > 1: the ranks take the MIN of their flags before the first update
        # of the step (GradReducer._agree_ok), so a batch that is poisoned on one rank is skipped by all.
        self.ok: Optional[Tensor] = torch.ones((), dtype=torch.float32, device=flat.flat_p.device)
        self.steps = 0
        self.runs: Optional[Dict[tuple, Tensor]] = None    # per bucket (s, e): the runs NOT updated in a GEMM epilogue
        self.skip: set = set()                             # ids of the parameters updated in a GEMM epilogue (those runs exclude)
        self.hyper: Optional[List[tuple]] = None           # per parameter (lr, wd); None = self.lr / self.wd for all
        self._class_runs: Dict[tuple, Tensor] = {}

    def flags(self) -> int:
        """AFFT_SGD_* flag word of the update kernels (include/afft_hip.h)"""
        return (1 if self.steps == 0 else 0) | (0 if self.nesterov else 2)

    def set_hyper(self, hyper: Optional[List[tuple]]):
        """one (lr, weight_decay) per parameter of the flat buffers, or None; collapses to the scalar form when all agree"""
        if hyper is not None:
            assert len(hyper) == len(self.flat.params)
            first = hyper[0]
            if all(h == first for h in hyper):
                self.lr, self.wd = float(first[0]), float(first[1])
                hyper = None
        self.hyper = hyper

    def hyper_of(self, i: int) -> tuple:
        return self.hyper[i] if self.hyper is not None else (self.lr, self.wd)

    def _runs_of(self, s: int, e: int, idx: tuple, skip_fused: bool) -> Tensor:
        """{start, length} runs (<= 16 Ki elements each: one block of the runs kernel) covering the parameters `idx` of bucket
        [s, e), without the ones updated in GEMM epilogues"""
        key = (s, e, idx, skip_fused)
        runs = self._class_runs.get(key)
        if runs is None:
            flat, CH = self.flat, 16384
            segs, cur = [], None
            for i in idx:
                p, o = flat.params[i], flat.offsets[i]
                if skip_fused and id(p) in self.skip:
                    cur = None
                    continue
                n = _align(p.numel())
                if cur is not None and cur[0] + cur[1] == o:
                    cur[1] += n
                else:
                    cur = [o, n]
                    segs.append(cur)
            chunks = [(a + k, min(CH, n - k)) for a, n in segs for k in range(0, n, CH)]
            runs = torch.tensor(chunks, dtype=torch.int64, device=flat.flat_p.device).reshape(-1, 2)
            if len(self._class_runs) > 4096:
                self._class_runs.clear()
            self._class_runs[key] = runs
        return runs

    def step_range(self, s: int, e: int, grad: Tensor, gscale: float, gscale_dev: Optional[Tensor] = None):
        fused = self.runs is not None and gscale_dev is None and grad.dtype == torch.float32
        if self.hyper is None:
            if fused:
                # the big weights of this bucket are updated in their weight-gradient epilogues (Trainer._enable_fused): one
                # launch over what is left of the bucket -- LayerNorm weights, biases, tokens, multiply-used weights
                runs = self.runs.get((s, e))
                if runs is not None:
                    if runs.shape[0]:
                        ops.sgd_nesterov_runs(self.flat.flat_p, self.flat.flat_g, self.buf, runs, self.lr, self.momentum, self.wd,
                                              gscale, self.flags(), p_bf16=self.flat.flat_p16, p_f16=self.flat.flat_h16, p_f8=self.flat.flat_p8, ok=self.ok)
                    self.flat.refresh_packed(s, e, skip=self.skip)
                    return
            p16 = self.flat.flat_p16[s:e] if self.flat.flat_p16 is not None else None
            ops.sgd_nesterov(self.flat.flat_p[s:e], grad, self.buf[s:e], self.lr, self.momentum, self.wd, gscale,
                             self.flags(), p_bf16=p16, gscale_dev=gscale_dev, p_f16=self.flat.h16(s, e), p_f8=self.flat.p8(s, e), ok=self.ok)
            self.flat.refresh_packed(s, e)
            return
        # per-parameter (lr, wd): the parameters of the range class by class.  [s, e) may cut a parameter at either end (the
        # sharded update hands over 1 / N of a bucket): every parameter is clipped to the range
        flat = self.flat
        classes: Dict[tuple, list] = {}
        for i, o in enumerate(flat.offsets):
            if o < e and o + _align(flat.params[i].numel()) > s:
                classes.setdefault(self.hyper[i], []).append(i)
        whole = all(s <= flat.offsets[i] and flat.offsets[i] + _align(flat.params[i].numel()) <= e for idx in classes.values() for i in idx)
        for (lr, wd), idx in classes.items():
            if grad.dtype == torch.float32 and gscale_dev is None and grad.numel() == e - s and whole:
                runs = self._runs_of(s, e, tuple(idx), fused)
                if runs.shape[0]:
                    # the runs kernel addresses the whole flat buffers: hand it the gradient at its flat position
                    g_full = flat.flat_g if grad.data_ptr() == flat.flat_g[s:e].data_ptr() else None
                    if g_full is not None:
                        ops.sgd_nesterov_runs(flat.flat_p, g_full, self.buf, runs, lr, self.momentum, wd, gscale, self.flags(),
                                              p_bf16=flat.flat_p16, p_f16=flat.flat_h16, p_f8=flat.flat_p8, ok=self.ok)
                        continue
                elif fused:
                    continue
            for i in idx:       # bf16 / clipped gradients, cut parameters: one launch per parameter (piece)
                p, o = flat.params[i], flat.offsets[i]
                if fused and id(p) in self.skip:
                    continue
                lo, hi = max(o, s), min(o + _align(p.numel()), e)
                p16 = flat.flat_p16[lo:hi] if flat.flat_p16 is not None else None
                ops.sgd_nesterov(flat.flat_p[lo:hi], grad[lo - s:hi - s], self.buf[lo:hi], lr, self.momentum, wd, gscale,
                                 self.flags(), p_bf16=p16, gscale_dev=gscale_dev, p_f16=flat.h16(lo, hi), p_f8=flat.p8(lo, hi), ok=self.ok)
        self.flat.refresh_packed(s, e, skip=(self.skip if fused else ()))

    def end_step(self):
        self.steps += 1
        rt.invalidate_weight_images()   # padded cast images (odd shapes) are re-cast on next use

    def step(self, grad: Optional[Tensor] = None, gscale: float = 1.0, grad_clip: Optional[float] = None):
        """grad_clip: clip by global L2 norm first (train.py:254-260 = torch.nn.utils.clip_grad_norm_): the norm, the
        coefficient min(1, c / (norm + 1e-6)) and its use all stay on the device (no host sync); the norm of the step
        is left in self.last_grad_norm (device scalar)."""
        g = self.flat.flat_g if grad is None else grad
        coef = None
        if grad_clip is not None and g.is_cuda:
            ss = torch.zeros(1, dtype=torch.float32, device=g.device)
            coef = torch.empty(1, dtype=torch.float32, device=g.device)
            self.last_grad_norm = torch.empty(1, dtype=torch.float32, device=g.device)
            ops.sumsq(g, ss, gscale * gscale)
            ops.clip_coef(ss, grad_clip, coef, self.last_grad_norm)
        self.step_range(0, self.flat.total, g, gscale, coef)
        self.end_step()


class _FusedEpilogue:
    """The optimizer fused into the weight-gradient GEMM epilogues (single GPU), shared by Trainer and afft_amd.optim.SGD.
    Needs: self.flat (FlatParams), self.opt (FusedSGD), self.reducer (GradReducer), self.grad_clip, self._fused (None until
    learned), self._name_of(p)."""

    _fused: Optional[Dict[int, object]] = None

    def _name_of(self, p: Tensor) -> str:
        return "?"

    # ---- optimizer fused into the weight-gradient GEMM epilogues (single GPU)
    def _can_fuse(self) -> bool:
        return (rt.fused_sgd() and rt.composite() and rt.precision() in ("bf16", "fp16x2") and rt.grad_mode() == "sink"
                and self.flat.flat_p.is_cuda and self.flat.flat_p16 is not None
                and not self.reducer.comm and self.grad_clip is None and not rt.CAPTURING)

    def _enable_fused(self):
        """After a step with the optimizer inside the backward pass: every GEMM weight that (a) went through a composite
        backward, (b) received exactly ONE gradient contribution and (c) owns a bf16 image in the flat buffers is from now on
        updated in the epilogue of its own weight-gradient GEMM (afft_sgd_fused_t): its gradient never goes to HBM and the
        per-bucket update kernel only walks what is left of the bucket (`runs`).  N = 1 only: with more ranks the summed
        gradient has to exist before the update."""
        from . import _lib as L_
        flat, fused = self.flat, {}
        for p, o in zip(flat.params, flat.offsets):
            img = getattr(p, "_afft_img", None)
            if (p.dim() == 2 and id(p) in rt.SINK.composite_weights and rt.SINK.touch_count.get(id(p), 0) == 1
                    and img is not None and img.external and p.numel() >= rt.fuse_min_elems()):
                d = L_.SgdFused()
                d.p, d.buf, d.p_bf16 = flat.flat_p.data_ptr() + 4 * o, self.opt.buf.data_ptr() + 4 * o, flat.flat_p16.data_ptr() + 2 * o
                fused[id(p)] = d
        self._fused = fused
        self.opt.runs = self._runs_without(fused)

    def _runs_without(self, fused) -> Dict[tuple, Tensor]:
        """per bucket: the {start, length} runs of the flat buffers that are NOT updated in a GEMM epilogue"""
        flat = self.flat
        CH = 16384       # a run is one 256-thread block of the runs kernel: keep them short
        self.opt.skip = set(fused)
        self.opt._class_runs.clear()
        runs = {}
        for (s, e) in self.reducer.buckets:
            segs, cur = [], None
            for p, o in zip(flat.params, flat.offsets):
                if not (s <= o < e):
                    continue
                n = _align(p.numel())
                if id(p) in fused:
                    cur = None
                    continue
                if cur is not None and cur[0] + cur[1] == o:
                    cur[1] += n
                else:
                    cur = [o, n]
                    segs.append(cur)
            chunks = [(a + k, min(CH, n - k)) for a, n in segs for k in range(0, n, CH)]
            runs[(s, e)] = torch.tensor(chunks, dtype=torch.int64, device=flat.flat_p.device).reshape(-1, 2)
        return runs

    def _audit_fused_step(self):
        """The set of epilogue-updated weights was learned on ONE step; a later step may route a weight differently (a branch
        that is skipped, a sub-layer shared by two call sites, the call-by-call path).  After every fused step, on the host:
        each such weight must have been updated in an epilogue exactly once and have had no other gradient contribution.
        * not updated in an epilogue this step: its whole gradient sits in the flat buffer (the first contribution
          overwrites, finish_step zeroes an untouched one) and the bucket kernel skipped it -> the regular update is applied
          to its range now (behind every bucket update: finish_step has joined the streams) and the weight leaves the set;
        * updated in an epilogue AND given a second contribution: the update already used a partial gradient -> error."""
        flat, sink, stale = self.flat, rt.SINK, []
        for p, o in zip(flat.params, flat.offsets):
            pid = id(p)
            if pid not in self._fused:
                continue
            applied, touches = sink.fused_applied.get(pid, 0), sink.touch_count.get(pid, 0)
            if applied == 1 and touches == 1:
                continue
            if applied != 0:
                raise RuntimeError(f"fused optimizer: weight {self._name_of(p)} was updated in its gradient GEMM's epilogue and then "
                                   f"received {touches - applied} more gradient contribution(s) in the same step (the graph changed "
                                   "since the fused set was learned).  Its parameter, momentum and bf16 image -- and every other "
                                   "bucket of this step -- HAVE ALREADY BEEN MODIFIED from a partial gradient: reload the last "
                                   "checkpoint, then rebuild the Trainer / optimizer or disable runtime.set_fused_sgd")
            n = _align(p.numel())
            lr, wd = self.opt.hyper_of(self._index[pid])
            ops.sgd_nesterov(flat.flat_p[o:o + n], flat.flat_g[o:o + n], self.opt.buf[o:o + n], lr, self.opt.momentum,
                             wd, 1.0, self.opt.flags(), p_bf16=flat.flat_p16[o:o + n], p_f16=flat.h16(o, o + n), p_f8=flat.p8(o, o + n), ok=self.opt.ok)
            flat.refresh_packed(o, o + 1)
            stale.append(pid)
        if stale:
            for pid in stale:
                del self._fused[pid]
            self.opt.runs = self._runs_without(self._fused)
            return True
        return False

    def _offset_of(self, p: Tensor) -> int:
        return self.flat.offsets[self._index[id(p)]]

    @property
    def _index(self) -> Dict[int, int]:
        ix = getattr(self, "_index_cache", None)
        if ix is None:
            ix = self._index_cache = self.flat.index_of()
        return ix

    def _fused_desc(self, p: Tensor):
        d = self._fused.get(id(p))
        if d is not None:
            lr, wd = self.opt.hyper_of(self._index[id(p)])
            d.p_pk16 = p._afft_img.pk.data_ptr() if rt.packed_live(p) else None      # only images a forward GEMM uses are kept fresh
            h16 = self.flat.flat_h16
            d.p_f16 = (h16.data_ptr() + 2 * self._offset_of(p)) if h16 is not None else None
            d.p_f8 = (self.flat.flat_p8.data_ptr() + self._offset_of(p)) if self.flat.flat_p8 is not None else None
            d.lr, d.mom, d.wd, d.gscale, d.first_step = lr, self.opt.momentum, wd, 1.0, self.opt.flags() & 2
            d.ok = self.opt.ok.data_ptr() if self.opt.ok is not None else None
        return d

class Trainer(_FusedEpilogue):
    """fwd + loss + bwd (+ overlapped gradient all-reduce) + fused SGD for a BaseModel.  With `overlap_optimizer`
    the update of a bucket runs on the side stream as soon as that bucket's gradient is final, under the
    backward GEMMs of the layers below it."""

    def _name_of(self, p: Tensor) -> str:
        return next((k for k, q in self.model.named_parameters() if q is p), "?")

    def __init__(self, model, loss_wts: Dict[str, float], lr=1e-3, momentum=0.9, weight_decay=1e-6,
                 comm_dtype: str = "fp32", bucket_elems: int = 32 * 1024 * 1024, group=None,
                 overlap_optimizer: bool = True, force_comm: bool = False, grad_clip: Optional[float] = None,
                 comm_algo: str = "allreduce"):
        from .common.runner import BasicLossAccuracy, Runner
        self.model = model
        self.flat = FlatParams(model, sharded_layout=(comm_algo == "sharded"))
        self.reducer = GradReducer(self.flat, group=group, bucket_elems=bucket_elems, comm_dtype=comm_dtype,
                                   force_comm=force_comm, comm_algo=comm_algo)
        self.opt = FusedSGD(self.flat, lr, momentum, weight_decay)
        self.reducer.opt_buf = self.opt.buf
        if comm_algo == "sharded":
            # whole fp32 masters before anybody reads them: refreshed at the first evaluation / no-grad forward after training steps
            # (every rank validates, train.py:399-401), checked -- never communicated -- by state_dict() (rank 0 alone saves)
            model.register_forward_pre_hook(lambda mod, *a: self._sync_if_evaluating(mod))
            model.register_state_dict_pre_hook(lambda *a, **k: self.reducer.assert_masters_fresh("state_dict()"))
        if self.reducer.world > 1:
            self.sync_parameters(group)
        self.loss_fn = BasicLossAccuracy(compute_metrics=False)
        self._reduce = Runner._reduce_loss
        self.loss_wts = loss_wts
        self.grad_clip = grad_clip     # opt.grad_clip of the reference's config; needs the whole gradient first
        self.overlap_optimizer = overlap_optimizer and grad_clip is None and (self.flat.flat_p.is_cuda or comm_algo == "sharded")
        self._fused: Optional[Dict[int, object]] = None    # id(weight) -> _lib.SgdFused, once learned (see _enable_fused)

    def sync_masters(self):
        """sharded update: whole fp32 masters and momentum on every rank (a collective: every rank calls it)"""
        self.reducer.sync_masters()

    def _sync_if_evaluating(self, mod):
        if self.reducer.masters_stale and (not mod.training or not torch.is_grad_enabled()):
            self.reducer.sync_masters()

    def sync_parameters(self, group=None, src: int = 0):
        """Every replica starts from rank `src`'s parameters and momentum (what torch DDP does at construction,
        train.py:364-368): only gradients are exchanged afterwards, so replicas that differ here never meet again --
        a checkpoint loaded on one rank, an unseeded init.  Refreshes the bf16 weight images from the received values.
        Runs at construction; call it again after loading a checkpoint into an existing Trainer."""
        if not (dist.is_available() and dist.is_initialized()):
            return
        root = dist.get_global_rank(group, src) if group is not None else src
        dist.broadcast(self.flat.flat_p, src=root, group=group)
        dist.broadcast(self.opt.buf, src=root, group=group)
        steps = torch.tensor([self.opt.steps], dtype=torch.int64, device=self.flat.flat_p.device)
        dist.broadcast(steps, src=root, group=group)
        self.opt.steps = int(steps)
        # the rest of the module state, as DDP broadcasts it: frozen parameters (not in the flat buffers) and buffers
        flat_ids = {id(p) for p in self.flat.params}
        for t in list(self.model.parameters()) + list(self.model.buffers()):
            if id(t) not in flat_ids:
                dist.broadcast(t.data, src=root, group=group)
        self.flat.refresh_images()

    def forward_backward(self, feats: Dict[str, Tensor], target, target_subclips, optimize_in_backward: bool = False,
                         mixup_fn: Optional[Callable] = None, mixup_backbone: bool = True):
        """mixup_fn / mixup_backbone: as in Runner.__call__ (common/runner.py:238-249; expts/01 trains with
        train.use_mixup=true, train.mixup_backbone=true): MixUp inside BaseModel.forward after the backbones, or on the
        features before the model; the losses then take soft targets and the ignore mask MixUp returns."""
        self.reducer.on_bucket = self.opt.step_range if optimize_in_backward else None
        if not optimize_in_backward and self.reducer.masters_stale:
            self.reducer.sync_masters()      # a replicated whole-buffer update follows: it must not start from stale masters / momentum
        fuse = optimize_in_backward and self._fused is not None and self._can_fuse()
        rt.SINK.fused = self._fused_desc if fuse else None
        rt.SINK.step_ok = self.reducer.step_ok = self.opt.ok      # see FusedSGD.ok; N > 1: GradReducer._agree_ok
        saved_runs, self.opt.runs = self.opt.runs, (self.opt.runs if fuse else None)
        try:
            self.reducer.begin_step()
            kwargs = dict(mixup_fn=None, target=target, target_subclips=target_subclips, target_subclips_ignore_index=None)
            if mixup_fn is not None:
                if mixup_backbone:
                    kwargs['mixup_fn'] = mixup_fn
                else:
                    feats, target, target_subclips, ign = mixup_fn(feats, target, target_subclips)
                    kwargs.update(target=target, target_subclips=target_subclips, target_subclips_ignore_index=ign)
            outputs, out_t = self.model(feats, **kwargs)
            losses, _ = self.loss_fn(outputs, out_t['target'], out_t['target_subclips'], mixup_enable=mixup_fn is not None,
                                     target_subclips_ignore_index=out_t['target_subclips_ignore_index'])
            loss, parts = self._reduce(losses, self.loss_wts, sync=False)
            one = getattr(self, "_one", None)      # the seed gradient, kept: autograd would allocate and fill a new one every step
            if one is None or one.device != loss.device or one.dtype != loss.dtype:
                one = self._one = torch.ones((), dtype=loss.dtype, device=loss.device)
            loss.backward(gradient=one)
            self.reducer.finish_step()
            if fuse and self._audit_fused_step():
                saved_runs = self.opt.runs      # the set shrank: keep the rebuilt runs
        finally:
            rt.SINK.fused = None
            rt.SINK.step_ok = None
            rt.SINK.fused_applied.clear()      # ids are only meaningful inside the step that recorded them
            self.opt.runs = saved_runs
        if optimize_in_backward:
            self.opt.end_step()
            if self._fused is None and self._can_fuse():
                self._enable_fused()
        return loss.detach(), parts

    # ---- captured step (hipGraph): for configurations whose step is bound by the host's enqueue rate
    def capture(self, feats, target, target_subclips, warmup: int = 3, single_stream: bool = False):
        """Capture one whole training step (forward, loss, backward with its three streams, fused SGD) on THESE input
        tensors into a hipGraph; step() then replays it whenever it is called with the same tensors (new batches are
        copied into them).  Single GPU only (the RCCL all-reduce is not captured).  The dropout masks still change every
        step: the keys drawn on the host are frozen in the graph, the salt they are XOR-ed with lives in device memory
        and is advanced by the first node of the graph.
        single_stream: capture the step as ONE linear chain (weight gradients on the main stream, the optimizer after the
        backward pass): no cross-stream edges in the graph, for the configurations whose eager step is bound by the host's
        enqueue rate rather than by the GPU."""
        from . import dropout as D_
        assert self.flat.flat_p.is_cuda and not self.reducer.comm, "capture(): single-GPU training only"
        assert single_stream or self.overlap_optimizer, "capture(): needs the optimizer inside the backward pass (no gradient clipping)"
        self._graph_single = bool(single_stream)
        D_.enable_device_salt(self.flat.flat_p.device)
        self._graph = None
        # warm up ON the capture stream: the library keeps its split-K workspace per stream and would otherwise
        # allocate it (hipMalloc) inside the capture; also learns the bucket counts and sets the kernel attributes
        cap = rt.new_stream(self.flat.flat_p.device, self.reducer.side_stream, self.reducer.opt_stream)
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            for _ in range(max(warmup, 2)):
                self._eager_step(feats, target, target_subclips)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        rt.CAPTURING = True
        try:
            with torch.cuda.graph(graph, stream=cap):
                loss, parts = self._eager_step(feats, target, target_subclips)
        finally:
            rt.CAPTURING = False
            rt.KEEPALIVE.clear()
        self._graph = graph
        # strong references to every tensor the captured kernels read: the replay reads THESE addresses, so a new batch
        # (features and labels alike) is copied into them, and they cannot be freed or recycled while the graph lives
        self._graph_in = (dict(feats), dict(target), dict(target_subclips) if target_subclips is not None else None)
        self._graph_io = (loss, parts)
        return self

    def release_graph(self):
        """back to eager steps (the device salt stays on: eager kernels read it too)"""
        self._graph = None
        self._graph_io = None
        self._graph_in = None

    def _eager_step(self, feats, target, target_subclips, mixup_fn=None, mixup_backbone=True):
        from . import dropout as D_
        D_.salt_step()
        if getattr(self, "_graph_single", False):
            was = rt.overlap_wgrad()
            rt.set_overlap_wgrad(False)
            try:
                loss, parts = self.forward_backward(feats, target, target_subclips, optimize_in_backward=False,
                                                    mixup_fn=mixup_fn, mixup_backbone=mixup_backbone)
                g, scale = self.reducer.grad_for_optimizer()
                self.opt.step(g, scale, grad_clip=self.grad_clip)
            finally:
                rt.set_overlap_wgrad(was)
            return loss, parts
        return self.forward_backward(feats, target, target_subclips, optimize_in_backward=True, mixup_fn=mixup_fn,
                                     mixup_backbone=mixup_backbone)

    def _feed_graph(self, feats, target, target_subclips) -> bool:
        """Copy a batch into the tensors the captured graph reads (no-op for the captured tensors themselves).  False when
        the batch does not have the captured structure (keys, shapes, dtypes): the caller then runs an eager step."""
        cf, ct, cs = self._graph_in
        pairs = []
        for cap, new in ((cf, feats), (ct, target), (cs, target_subclips)):
            if (cap is None) != (new is None):
                return False
            if cap is None:
                continue
            if cap.keys() != new.keys():
                return False
            for k, t in cap.items():
                n = new[k]
                if n.shape != t.shape or n.dtype != t.dtype:
                    return False
                pairs.append((t, n))
        for t, n in pairs:
            if n is not t:
                t.copy_(n, non_blocking=True)
        return True

    def step(self, feats, target, target_subclips, optimize: bool = True, mixup_fn: Optional[Callable] = None,
             mixup_backbone: bool = True):
        g = getattr(self, "_graph", None)
        if g is not None and optimize and mixup_fn is None and self._feed_graph(feats, target, target_subclips):
            g.replay()
            return self._graph_io
        fused = optimize and self.overlap_optimizer
        loss, parts = self.forward_backward(feats, target, target_subclips, optimize_in_backward=fused, mixup_fn=mixup_fn,
                                            mixup_backbone=mixup_backbone)
        if optimize and not fused:
            g, scale = self.reducer.grad_for_optimizer()
            self.opt.step(g, scale, grad_clip=self.grad_clip)
        return loss, parts


class DistributedDataParallel(torch.nn.Module):
    """``torch.nn.parallel.DistributedDataParallel``-shaped wrapper for the reference's ``train.py:364-368``::

        model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[gpu], output_device=gpu)

    (``afft_amd.install_as_models(patch_ddp=True)`` makes that very line construct this class.)  torch's wrapper learns about a
    gradient from an autograd hook on the parameter; on the default "sink" path the weight-gradient GEMMs write (and, on one GPU,
    consume) gradients without autograd ever seeing them, so torch DDP would wait for gradients that never arrive.  This wrapper
    * broadcasts rank 0's parameters and buffers at construction (and, when the parameters belong to an ``afft_amd.optim.SGD``,
      its momentum and step count; the bf16 weight images are re-derived) -- what torch DDP's constructor does;
    * with an ``afft_amd.optim.SGD`` over the same parameters does nothing else: that optimizer's ``GradReducer`` already
      all-reduces bucket by bucket during backward (RCCL, side stream) and applies 1 / world in its update;
    * with any other optimizer (``torch.optim.SGD`` ...) owns a ``FlatParams`` + ``GradReducer`` itself: ``forward`` (training
      mode) opens the step, a hook on the outputs queues an end-of-backward callback that completes the all-reduce, divides by
      the world size and joins the streams, so ``optimizer.step()`` sees averaged gradients in ``p.grad`` as under torch DDP.
    ``state_dict()`` keys carry the same ``module.`` prefix as torch DDP's."""

    def __init__(self, module: torch.nn.Module, device_ids=None, output_device=None, dim: int = 0, broadcast_buffers: bool = True,
                 process_group=None, bucket_cap_mb: Optional[float] = None, comm_dtype: str = "fp32", comm_algo: str = "allreduce",
                 **unused):
        super().__init__()
        self.module = module
        self.device_ids, self.output_device, self.dim = device_ids, output_device, dim
        self.process_group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        from . import optim as _optim
        self._engines = _optim.engines_for(module)
        self._own = None
        self._pending = False
        if self.world > 1:
            root = dist.get_global_rank(process_group, 0) if process_group is not None else 0
            owned = set()
            for e in self._engines:
                e.sync_parameters(process_group)
                owned |= {id(p) for p in e.flat.params}
            with torch.no_grad():
                for t in list(module.parameters()) + (list(module.buffers()) if broadcast_buffers else []):
                    if id(t) not in owned:
                        dist.broadcast(t.data, src=root, group=process_group)
            rt.invalidate_weight_images()
        if not self._engines and any(p.requires_grad for p in module.parameters()):
            flat = FlatParams(module)
            elems = 32 * 1024 * 1024 if bucket_cap_mb is None else max(1, int(bucket_cap_mb * 2 ** 20 / 4))
            red = GradReducer(flat, group=process_group, bucket_elems=elems, comm_dtype=comm_dtype, comm_algo=comm_algo)
            red.defer_all = True      # a foreign zero_grad(set_to_none=True) between forward and backward detaches p.grad from the
            self._own = (flat, red)   # flat buffer: gradients are gathered into it when backward is over, then reduced

    def state_dict(self, *args, **kwargs):
        for e in self._engines:      # sharded update: whole fp32 masters before they are read -- checked, not communicated (rank 0 alone saves)
            e.reducer.assert_masters_fresh("DistributedDataParallel.state_dict()")
        return super().state_dict(*args, **kwargs)

    # ---- foreign optimizer: the wrapper brackets the backward pass itself
    def _begin(self):
        flat, red = self._own
        for p, o in zip(flat.params, flat.offsets):      # a torch optimizer's zero_grad(set_to_none=True) dropped the views
            if p.grad is None or p.grad.data_ptr() != flat.flat_g.data_ptr() + 4 * o:
                if rt.grad_mode() != "sink":
                    flat.flat_g[o:o + p.numel()].zero_()
                p.grad = flat.flat_g[o:o + p.numel()].view(p.shape)
        red.on_bucket = None
        red.begin_step()
        self._pending = True

    def _finish(self):
        if not self._pending:
            return
        self._pending = False
        flat, red = self._own
        if flat.flat_g.is_cuda and rt.overlap_wgrad():      # weight gradients were enqueued on the auxiliary stream
            torch.cuda.current_stream().wait_stream(rt.aux_stream(flat.flat_g.device))
        with torch.no_grad():
            for p, o in zip(flat.params, flat.offsets):     # gradients produced into fresh tensors (p.grad was None): gather them
                view = flat.flat_g[o:o + p.numel()].view(p.shape)
                if p.grad is None:
                    view.zero_()
                elif p.grad.data_ptr() != view.data_ptr():
                    view.copy_(p.grad)
                rt.SINK.touched[id(p)] = True
                p.grad = view
        red.finish_step()
        if red.comm:
            g, scale = red.grad_for_optimizer()
            if g is not flat.flat_g:                      # bf16 payload: back into the fp32 gradient buffer
                flat.flat_g.copy_(g)
            flat.flat_g.mul_(scale)

    def _arm_outputs(self, out):
        def visit(o):
            if isinstance(o, torch.Tensor):
                if o.requires_grad:
                    o.register_hook(self._on_first_grad)
            elif isinstance(o, dict):
                for v in o.values():
                    visit(v)
            elif isinstance(o, (list, tuple)):
                for v in o:
                    visit(v)
        visit(out)

    def _on_first_grad(self, grad):
        if self._pending and not getattr(self, "_queued", False):
            self._queued = True

            def _final():
                self._queued = False
                self._finish()
            torch.autograd.Variable._execution_engine.queue_callback(_final)
        return grad

    def forward(self, *args, **kwargs):
        bracket = self._own is not None and self.training and torch.is_grad_enabled()
        if not (self.training and torch.is_grad_enabled()):
            for e in self._engines:      # sharded update: the first evaluation forward after training steps refreshes the masters (every rank validates)
                e.sync_masters()
        if bracket:
            self._begin()
        out = self.module(*args, **kwargs)
        if bracket:
            self._arm_outputs(out)
        return out

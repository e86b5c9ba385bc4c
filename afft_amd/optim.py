"""``afft_amd.optim.SGD``: the flat-buffer fused SGD of ``afft_amd.parallel`` behind the ``torch.optim.Optimizer`` interface, so
that the reference's own training loop reaches the fast path unchanged.

The reference builds its optimizer with ``hydra.utils.instantiate(cfg.opt.optimizer, param_groups)`` over the 151 per-parameter
groups of ``prepare_params`` (train.py:189-225, :352; conf/opt/optimizer/sgd.yaml: ``_target_: torch.optim.SGD``, momentum 0.9,
``nesterov`` per experiment) and then runs (train.py:228-265)::

    loss, metrics = runner(data, mixup_fn, mixup_backbone)
    optimizer.zero_grad(); loss.backward(); [clip_grad_norm_]; optimizer.step(); lr_scheduler.step()

Selecting this class is one Hydra override, no file of the reference changes: ``opt.optimizer._target_=afft_amd.optim.SGD``.

What the five calls become here:

* construction: every parameter of the groups (and its ``.grad``) is re-homed in the flat fp32 buffers of
  ``parallel.FlatParams`` (one bf16 image per GEMM weight at the same offsets), a ``parallel.GradReducer`` is set up over them
  (``torch.distributed`` already initialised and more than one rank: bucketed all-reduce over RCCL, overlapped with backward);
* ``zero_grad()`` is the start of a backward pass: nothing is zeroed (the first weight-gradient GEMM of a step overwrites, later
  ones accumulate, untouched gradients are zeroed at the end -- ``runtime.GradSink``); the learning rate and weight decay of
  every group are read HERE (so ``common/scheduler.py``'s ``Warmup`` / ``CosineLR``, which write ``param_groups[i]['lr']`` in
  ``lr_scheduler.step()`` after ``optimizer.step()``, drive the kernels exactly as they drive ``torch.optim.SGD``), and the
  per-bucket update is armed to run INSIDE the backward pass -- in the epilogue of each weight-gradient GEMM on one GPU
  (``afft_sgd_fused_t``), on the optimizer stream behind each bucket's all-reduce with more ranks;
* ``loss.backward()`` then is the whole step;
* ``step()`` joins the streams, audits the fused set, counts the step and publishes the momentum buffers as
  ``state[p]['momentum_buffer']`` (views of the flat buffer: ``state_dict()`` / ``load_state_dict()`` round-trip through
  ``torch.save`` like torch's own, train.py:161-176);
* gradient clipping (``opt.grad_clip``, train.py:254-260) needs the whole gradient before the first update: construct with
  ``in_backward=False`` (Hydra: ``+opt.optimizer.in_backward=false``) or set ``AFFT_OPT_IN_BACKWARD=0``; the gradient exchange is
  then completed -- all-reduced, averaged over the ranks, untouched gradients zeroed -- by a callback at the END of
  ``loss.backward()``, so the loop's ``clip_grad_norm_`` sees the complete, averaged flat gradient on any number of ranks, and
  ``step()`` runs the one-launch-per-bucket update.

Under ``AFFT_GRAD_MODE=autograd`` (gradients through autograd, e.g. below torch's own DistributedDataParallel, which then owns
the all-reduce) ``zero_grad()`` zeroes the flat gradient buffer with one fill and ``step()`` is the plain fused update over it.
"""
from __future__ import annotations

import os
import weakref
from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from . import ops, runtime as rt
from .parallel import FlatParams, FusedSGD, GradReducer, _FusedEpilogue

Tensor = torch.Tensor

_ENGINES: List["weakref.ref"] = []     # live SGD instances (parallel.DistributedDataParallel looks its optimizer up here)


def engines_for(module: torch.nn.Module) -> List["SGD"]:
    """the afft optimizers whose parameters all belong to `module`"""
    ids = {id(p) for p in module.parameters()}
    out = []
    for r in list(_ENGINES):
        o = r()
        if o is None:
            _ENGINES.remove(r)
        elif all(id(p) in ids for p in o.flat.params):
            out.append(o)
    return out


class SGD(torch.optim.Optimizer, _FusedEpilogue):
    """torch.optim.SGD(params, lr, momentum, dampening=0, weight_decay, nesterov) on the flat fused path (module docstring).
    Extra keyword arguments: comm_dtype ('fp32' | 'bf16' gradient payload), comm_algo ('allreduce' | 'rs_ag' | 'sharded': the
    update of the GEMM weights in 1 / N slices with an all-gather of their 16-bit images, parallel.GradReducer), bucket_elems,
    group, in_backward (update inside the backward pass; default on), grad_clip (clip by global norm inside step(), on the
    device, instead of the loop's clip_grad_norm_ call)."""

    def __init__(self, params, lr: float = 1e-3, momentum: float = 0.0, dampening: float = 0.0, weight_decay: float = 0.0,
                 nesterov: bool = False, *, maximize: bool = False, foreach=None, differentiable: bool = False, fused=None,
                 comm_dtype: Optional[str] = None, comm_algo: Optional[str] = None, bucket_elems: int = 32 * 1024 * 1024,
                 group=None, in_backward: Optional[bool] = None, grad_clip: Optional[float] = None):
        if lr < 0.0 or momentum < 0.0 or weight_decay < 0.0:
            raise ValueError("afft_amd.optim.SGD: negative lr / momentum / weight_decay")
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")      # torch's own check
        if maximize or differentiable:
            raise NotImplementedError("afft_amd.optim.SGD: maximize / differentiable are not supported")
        defaults = dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov)
        super().__init__(params, defaults)
        g0 = self.param_groups[0]
        for g in self.param_groups:
            if g["dampening"] != 0:
                raise NotImplementedError("afft_amd.optim.SGD: dampening != 0 is not supported (the reference never sets it)")
            if g["momentum"] != g0["momentum"] or g["nesterov"] != g0["nesterov"]:
                raise NotImplementedError("afft_amd.optim.SGD: momentum / nesterov must be the same in every group "
                                          "(train.py:189-225 varies lr and weight_decay only)")
        plist = [p for g in self.param_groups for p in g["params"]]
        algo = comm_algo or os.environ.get("AFFT_COMM_ALGO", "allreduce")
        self.flat = FlatParams(plist, sharded_layout=(algo == "sharded"))
        if len(self.flat.params) != len(plist):
            raise ValueError("afft_amd.optim.SGD: every parameter must require grad and appear once (train.py:219-224 drops the "
                             "lr = 0 groups and clears requires_grad on their parameters)")
        own_comm = rt.grad_mode() == "sink"       # autograd mode: whoever wraps the model (torch DDP) reduces
        self.reducer = GradReducer(self.flat, group=group, bucket_elems=bucket_elems,
                                   comm_dtype=comm_dtype or os.environ.get("AFFT_COMM_DTYPE", "fp32"),
                                   comm_algo=algo)
        if not own_comm:
            self.reducer.comm = False
        self.opt = FusedSGD(self.flat, lr, g0["momentum"], weight_decay, nesterov=bool(g0["nesterov"]))
        self.grad_clip = grad_clip
        if in_backward is None:
            in_backward = os.environ.get("AFFT_OPT_IN_BACKWARD", "1") != "0"
        self.in_backward = bool(in_backward) and grad_clip is None and (self.flat.flat_p.is_cuda or algo == "sharded")
        self._fused = None
        self._armed = False
        self._fuse_now = False
        self._saved_runs = None
        self._names: Dict[int, str] = {id(p): g.get("name", "?") for g in self.param_groups for p in g["params"]}
        self._group_of: List[dict] = self._groups_in_flat_order()
        self.reducer.opt_buf = self.opt.buf
        if self.reducer.world > 1 and self.reducer.comm:
            self.sync_parameters(group)
        _ENGINES.append(weakref.ref(self))

    # ------------------------------------------------------------------ helpers
    def _groups_in_flat_order(self) -> List[dict]:
        """the parameter group of every parameter of the flat buffers, in THEIR order (the sharded layout moves the GEMM weights first)"""
        gmap = {id(p): g for g in self.param_groups for p in g["params"]}
        return [gmap[id(p)] for p in self.flat.params]

    def sync_masters(self):
        """sharded update (comm_algo = 'sharded'): whole fp32 masters and momentum on every rank (a collective: EVERY rank calls it;
        parallel.DistributedDataParallel does at the first evaluation forward after training steps)"""
        self.reducer.sync_masters()

    def state_dict(self):
        self.reducer.assert_masters_fresh("afft_amd.optim.SGD.state_dict()")      # never a collective: rank 0 alone saves (train.py:403-411)
        return super().state_dict()

    def _name_of(self, p: Tensor) -> str:
        return self._names.get(id(p), "?")

    def _sync_hyper(self):
        """param_groups -> the kernels' (lr, weight_decay): read once per step (a scheduler rewrites group['lr'] every iteration)"""
        self.opt.set_hyper([(float(g["lr"]), float(g["weight_decay"])) for g in self._group_of])

    def sync_parameters(self, group=None, src: int = 0):
        """every replica starts from rank `src`'s parameters, momentum and step count (what torch DDP's constructor does for the
        parameters, train.py:364-368); the bf16 images are re-derived from the received values"""
        if not (dist.is_available() and dist.is_initialized()):
            return
        root = dist.get_global_rank(group, src) if group is not None else src
        dist.broadcast(self.flat.flat_p, src=root, group=group)
        dist.broadcast(self.opt.buf, src=root, group=group)
        steps = torch.tensor([self.opt.steps], dtype=torch.int64, device=self.flat.flat_p.device)
        dist.broadcast(steps, src=root, group=group)
        self.opt.steps = int(steps)
        self.flat.refresh_images()

    # ------------------------------------------------------------------ the torch.optim.Optimizer interface
    def zero_grad(self, set_to_none: bool = True):
        """Start of a backward pass (module docstring).  set_to_none is ignored: the gradients live in the flat buffer."""
        if self._armed:
            self._disarm()
        if rt.grad_mode() != "sink":
            self.flat.flat_g.zero_()
            for p, o in zip(self.flat.params, self.flat.offsets):     # a foreign zero_grad(set_to_none=True) may have dropped the views
                if p.grad is None:
                    p.grad = self.flat.flat_g[o:o + p.numel()].view(p.shape)
            return
        self._sync_hyper()
        inb = self.in_backward
        self.reducer.on_bucket = self.opt.step_range if inb else None
        if not inb and self.reducer.masters_stale:
            self.reducer.sync_masters()      # a replicated whole-buffer update follows (every rank takes this branch): not from stale masters
        self._fuse_now = inb and self._fused is not None and self._can_fuse()
        rt.SINK.fused = self._fused_desc if self._fuse_now else None
        rt.SINK.step_ok = self.reducer.step_ok = self.opt.ok      # parallel.FusedSGD.ok: a non-finite loss makes the step a no-op (all ranks agree)
        self._saved_runs, self.opt.runs = self.opt.runs, (self.opt.runs if self._fuse_now else None)
        self.reducer.begin_step()
        self._armed = True
        self._finished_in_backward = False
        if not inb:
            # The update waits for step(), but the GRADIENT has to be whole when backward() returns: the loop's own
            # torch.nn.utils.clip_grad_norm_ (train.py:254-260) reads and scales p.grad between the two calls.  The first gradient
            # that reports ready queues an end-of-backward callback that completes the exchange (joins the side streams, zeroes
            # untouched gradients) and applies 1 / world, as parallel.DistributedDataParallel does for a foreign optimizer.
            inner = rt.SINK.on_grad_ready
            self._end_queued = False

            def first_ready(p, inner=inner):
                if not self._end_queued:
                    try:
                        torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)
                        self._end_queued = True
                    except RuntimeError:      # not inside a backward pass (a Function.backward called by hand): step() finishes
                        pass
                if inner is not None:
                    inner(p)
            rt.SINK.on_grad_ready = first_ready

    def _end_of_backward(self):
        """in_backward = False: runs when the backward pass is over (autograd-engine callback)"""
        self._end_queued = False
        if not self._armed or self._finished_in_backward:
            return
        red = self.reducer
        red.finish_step()
        if red.comm:
            g, scale = red.grad_for_optimizer()
            with torch.no_grad():
                if g is not self.flat.flat_g:       # bf16 payload: back into the fp32 gradient buffer the loop (and step()) read
                    self.flat.flat_g.copy_(g)
                self.flat.flat_g.mul_(scale)
        self._finished_in_backward = True

    def _disarm(self):
        rt.SINK.fused = None
        rt.SINK.step_ok = None
        rt.SINK.fused_applied.clear()      # ids are only meaningful inside the step that recorded them
        rt.SINK.on_grad_ready = None
        self.opt.runs = self._saved_runs
        self._armed = False

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            raise NotImplementedError("afft_amd.optim.SGD.step: closures are not supported (the update runs inside backward())")
        if self._armed:
            done = getattr(self, "_finished_in_backward", False)
            try:
                if not done:
                    self.reducer.finish_step()
                if self._fuse_now and self._audit_fused_step():
                    self._saved_runs = self.opt.runs          # the fused set shrank: keep the rebuilt runs
            finally:
                self._disarm()
            if self.in_backward:
                self.opt.end_step()
                if self._fused is None and self._can_fuse():
                    self._enable_fused()
            elif done:      # the exchange was completed and averaged when backward() ended (and the loop may have clipped since)
                self.opt.step(self.flat.flat_g, 1.0, grad_clip=self.grad_clip)
            else:
                g, scale = self.reducer.grad_for_optimizer()
                self.opt.step(g, scale, grad_clip=self.grad_clip)
        else:
            # gradients arrived through autograd (AFFT_GRAD_MODE=autograd, e.g. below torch DDP, which has averaged them), or the
            # caller accumulated several backward passes without zero_grad(): the plain update over the whole flat gradient
            if self.flat.flat_g.is_cuda and rt.overlap_wgrad():
                torch.cuda.current_stream().wait_stream(rt.aux_stream(self.flat.flat_g.device))
            self._sync_hyper()
            self.opt.step(self.flat.flat_g, 1.0, grad_clip=self.grad_clip)
        self._publish_state()
        return loss

    # ------------------------------------------------------------------ optimizer state <-> flat momentum buffer
    def _publish_state(self):
        if getattr(self, "_state_published", False):
            return
        for p, o in zip(self.flat.params, self.flat.offsets):
            self.state[p]["momentum_buffer"] = self.opt.buf[o:o + p.numel()].view(p.shape)
        self._state_published = True

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._group_of = self._groups_in_flat_order()
        any_buf = False
        with torch.no_grad():
            for p, o in zip(self.flat.params, self.flat.offsets):
                st = self.state.get(p, {})
                mb = st.get("momentum_buffer")
                view = self.opt.buf[o:o + p.numel()].view(p.shape)
                if mb is not None:
                    any_buf = True
                    if mb.data_ptr() != view.data_ptr():
                        view.copy_(mb)
                self.state[p]["momentum_buffer"] = view
        self._state_published = True
        if any_buf:
            self.opt.steps = max(self.opt.steps, 1)      # the buffers exist: the next step is not a "first step"
        # the parameters themselves are loaded by the caller (model.load_state_dict writes through the flat views in place)
        self.flat.refresh_images()

    def add_param_group(self, param_group):
        if hasattr(self, "flat"):
            raise NotImplementedError("afft_amd.optim.SGD: parameter groups are fixed at construction (flat buffers)")
        super().add_param_group(param_group)

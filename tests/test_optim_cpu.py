"""Host logic of the drop-in optimizer / DDP-shaped wrapper / restated schedulers, in the build container: the
kernels are replaced by the torch test double (tests/cpu_ops.py), everything else is the product code.

  * afft_amd.optim.SGD in the reference's loop shape (train.py:228-265: Runner -> zero_grad -> backward -> step -> scheduler)
    == afft_amd.parallel.Trainer.step, bit for bit, and == torch.optim.SGD over the same per-parameter groups (to rounding);
  * Warmup(CosineLR) (common/scheduler.py:57-160 semantics) changes the learning rate the update kernel is called with;
  * per-module lr / weight_decay groups (train.py:199-212), nesterov on and off, state_dict round trip;
  * world_size 2 over gloo: the unchanged loop with afft_amd.optim.SGD + afft_amd.parallel.DistributedDataParallel equals a
    single process on the full batch; the wrapper alone (torch.optim.SGD) averages gradients like torch DDP.
"""
import copy
import math
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from test_parallel_cpu import WTS, _afft_case, _afft_model


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batch(data, tgt, sub, sl=slice(None)):
    return ({"data_dict": {m: d[sl] for m, d in data.items()}, "target": {"action": tgt[sl]},
             "target_subclips": {"action": sub[sl]}}, {})


def _loop(model, opt, sched, data, tgt, sub, steps, lrs=None, sl=slice(None), mixup_fn=None, clip=None, norms=None):
    """the reference's loop body (train.py:241-265); clip: opt.grad_clip, applied by the LOOP as the reference does (:254-260)"""
    from afft_amd.common.runner import Runner
    runner = Runner(model, torch.device("cpu"), WTS, compute_metrics=False)
    for _ in range(steps):
        loss, _metrics = runner(_batch(data, tgt, sub, sl), mixup_fn, True)
        opt.zero_grad()
        loss.backward()
        if clip is not None:
            n = torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
            if norms is not None:
                norms.append(float(n))
        opt.step()
        if sched is not None:
            sched.step()
        if lrs is not None:
            lrs.append(opt.param_groups[0]["lr"])
    return loss


def _groups(model, lr=1e-2, wd=1e-4, lr_wd=None):
    from afft_amd.common.scheduler import prepare_params
    return prepare_params(model, lr_wd, lr, wd)


def test_dropin_sgd_equals_trainer_and_torch_sgd():
    import cpu_ops
    from afft_amd.optim import SGD
    from afft_amd.parallel import Trainer
    c, state, data, tgt, sub = _afft_case()
    with cpu_ops.installed():
        m1 = _afft_model(c, state, "fp32")
        tr = Trainer(m1, WTS, lr=1e-2, momentum=0.9, weight_decay=1e-4, bucket_elems=8192)
        for _ in range(3):
            tr.step(data, {"action": tgt}, {"action": sub})
        m2 = _afft_model(c, state, "fp32")
        groups = _groups(m2)
        assert len(groups) == len(list(m2.parameters())) > 50           # one group per parameter, as prepare_params builds them
        opt = SGD(groups, lr=1e-2, momentum=0.9, nesterov=True, bucket_elems=8192)
        assert len(opt.reducer.buckets) >= 3
        _loop(m2, opt, None, data, tgt, sub, 3)
        assert torch.equal(opt.flat.flat_p, tr.flat.flat_p)             # the same kernels in the same order: bitwise
        assert torch.equal(opt.opt.buf, tr.opt.buf)
        # torch.optim.SGD on the same groups, gradients through the same sink
        m3 = _afft_model(c, state, "fp32")
        ref = torch.optim.SGD(_groups(m3), lr=1e-2, momentum=0.9, nesterov=True)
        _loop(m3, ref, None, data, tgt, sub, 3)
        for (n, p), q in zip(m2.named_parameters(), m3.parameters()):
            assert float((p.detach() - q.detach()).norm() / (q.detach().norm() + 1e-12)) < 1e-6, n
        # the momentum buffers are exposed the way torch's are
        sd = opt.state_dict()
        assert len(sd["state"]) == len(groups) and all("momentum_buffer" in s for s in sd["state"].values())
        for p, q in zip(m2.parameters(), m3.parameters()):
            assert torch.allclose(opt.state[p]["momentum_buffer"], ref.state[q]["momentum_buffer"], rtol=1e-4, atol=1e-6)


def test_schedulers_drive_the_kernel_learning_rate():
    """Warmup(CosineLR) per iteration: closed-form values, and the lr the update kernel is CALLED with follows them"""
    import cpu_ops
    from afft_amd import ops
    from afft_amd.common.scheduler import CosineLR, Warmup
    from afft_amd.optim import SGD
    c, state, data, tgt, sub = _afft_case()
    with cpu_ops.installed():
        model = _afft_model(c, state, "fp32")
        opt = SGD(_groups(model, lr=1e-2), lr=1e-2, momentum=0.9, nesterov=True, bucket_elems=1 << 30)
        ipe, wu, cos = 2, 2, 3         # iterations per epoch, warm-up epochs, cosine epochs
        sched = Warmup(opt, CosineLR(opt, num_epochs=cos, iters_per_epoch=ipe, world_size=2, eta_min=1e-6),
                       init_lr_ratio=0.01, num_epochs=wu, iters_per_epoch=ipe, world_size=2)
        seen = []
        real = ops.sgd_nesterov

        def spy(p, g, buf, lr, *a, **k):
            seen.append(lr)
            return real(p, g, buf, lr, *a, **k)
        ops.sgd_nesterov = spy
        try:
            lrs = [opt.param_groups[0]["lr"]]
            _loop(model, opt, sched, data, tgt, sub, 11, lrs)
        finally:
            ops.sgd_nesterov = real
    base, W, T, eta = 1e-2, wu * ipe, cos * ipe, 1e-6 * 2
    expect = [base * (0.01 + 0.99 * t / W) for t in range(W)]                                   # warm-up iterations 0..W-1
    expect += [eta + (base - eta) * 0.5 * (1 + math.cos(math.pi * t / T)) for t in range(1, T)]  # cosine picks up at its step 1
    expect += [0.0] * 4                                                                          # past T_max (scheduler.py:69-76)
    assert len(lrs) == 12
    for t, (a, b) in enumerate(zip(lrs, expect)):
        assert abs(a - b) <= 1e-12 + 1e-9 * abs(b), (t, a, b)
    # one kernel call per step (one bucket, uniform groups): step t ran with the lr set by scheduler step t - 1
    assert len(seen) == 11 and all(abs(a - b) < 1e-15 for a, b in zip(seen, lrs[:11])), (seen, lrs)


@pytest.mark.parametrize("nesterov", [True, False])
def test_per_module_lr_wd_groups_match_torch(nesterov):
    """train.py:199-212: `opt.lr_wd` gives sub-modules their own lr / weight decay; lr = 0 freezes"""
    import cpu_ops
    from afft_amd.optim import SGD
    c, state, data, tgt, sub = _afft_case()
    lr_wd = [[["future_predictor.future_predictor"], 3e-3, 0.0], [["future_predictor.classifiers"], 0.0, 0.0]]
    with cpu_ops.installed():
        m1, m2 = _afft_model(c, state, "fp32"), _afft_model(c, state, "fp32")
        names1 = [n for n, _ in m1.named_parameters()]
        if not any(n.startswith("future_predictor.future_predictor") for n in names1):
            pytest.skip("module names differ")
        g1, g2 = _groups(m1, lr_wd=lr_wd), _groups(m2, lr_wd=lr_wd)
        frozen = [n for n, p in m1.named_parameters() if not p.requires_grad]
        assert frozen and all(n.startswith("future_predictor.classifiers") for n in frozen)
        assert len({(g["lr"], g["weight_decay"]) for g in g1}) == 2
        opt = SGD(g1, lr=1e-2, momentum=0.9, nesterov=nesterov, bucket_elems=8192)
        ref = torch.optim.SGD(g2, lr=1e-2, momentum=0.9, nesterov=nesterov)
        _loop(m1, opt, None, data, tgt, sub, 3)
        _loop(m2, ref, None, data, tgt, sub, 3)
        assert opt.opt.hyper is not None            # really the per-parameter path
        for (n, p), q in zip(m1.named_parameters(), m2.parameters()):
            assert float((p.detach() - q.detach()).norm() / (q.detach().norm() + 1e-12)) < 1e-6, n


def test_state_dict_round_trip_and_accumulation_without_zero_grad():
    import cpu_ops
    from afft_amd.optim import SGD
    c, state, data, tgt, sub = _afft_case()
    with cpu_ops.installed():
        m1 = _afft_model(c, state, "fp32")
        o1 = SGD(_groups(m1), lr=1e-2, momentum=0.9, nesterov=True, bucket_elems=8192)
        _loop(m1, o1, None, data, tgt, sub, 2)
        sd_model = {k: v.clone() for k, v in m1.state_dict().items()}
        sd_opt = copy.deepcopy(o1.state_dict())        # state_dict() hands out references (views of the flat buffer), as torch's does
        _loop(m1, o1, None, data, tgt, sub, 2)
        # resume in a fresh model / optimizer from the checkpoint taken after 2 steps
        m2 = _afft_model(c, state, "fp32")
        o2 = SGD(_groups(m2), lr=1e-2, momentum=0.9, nesterov=True, bucket_elems=8192)
        m2.load_state_dict(sd_model)
        o2.load_state_dict(sd_opt)
        assert o2.opt.steps >= 1
        _loop(m2, o2, None, data, tgt, sub, 2)
        assert torch.equal(o2.flat.flat_p, o1.flat.flat_p) and torch.equal(o2.opt.buf, o1.opt.buf)


def test_trainer_step_with_mixup_matches_runner_loss():
    """Trainer.step(..., mixup_fn=) runs the recipe expts/01 trains (MixUp inside BaseModel.forward, soft-target losses)"""
    import cpu_ops
    from afft_amd.common.runner import Runner
    from afft_amd.parallel import Trainer
    c, state, data, tgt, sub = _afft_case()

    class FixedMix(torch.nn.Module):          # a deterministic stand-in with MixUp's return contract (common/mixup.py:119-182)
        def forward(self, x, labels, labels_subclips):
            K = c["num_classes"]
            lam = 0.7
            x2 = {m: lam * v + (1 - lam) * v.flip(0) for m, v in x.items()}
            oh = lambda t: torch.nn.functional.one_hot(t.squeeze(-1), K).float()    # noqa: E731
            lab = {k: lam * oh(v) + (1 - lam) * oh(v).flip(0) for k, v in labels.items()}
            sub2 = {k: lam * oh(v) + (1 - lam) * oh(v).flip(0) for k, v in labels_subclips.items()}
            ign = {k: torch.zeros(v.shape, dtype=torch.bool) for k, v in labels_subclips.items()}
            return x2, lab, sub2, ign

    with cpu_ops.installed():
        m1, m2 = _afft_model(c, state, "fp32"), _afft_model(c, state, "fp32")
        tr = Trainer(m1, WTS, lr=1e-2, momentum=0.9, weight_decay=1e-4, bucket_elems=8192)
        loss_t, _ = tr.step(data, {"action": tgt}, {"action": sub}, mixup_fn=FixedMix())
        runner = Runner(m2, torch.device("cpu"), WTS, compute_metrics=False)
        loss_r, _ = runner(_batch(data, tgt, sub), FixedMix(), True)
        assert abs(float(loss_t) - float(loss_r.detach())) < 1e-6 * abs(float(loss_r.detach()))
        plain, _ = Runner(m2, torch.device("cpu"), WTS, compute_metrics=False)(_batch(data, tgt, sub), None, True)
        assert abs(float(plain.detach()) - float(loss_r.detach())) > 1e-3           # the soft-target path really differs from the hard-label one


# ----------------------------------------------------------------------------- world_size 2 over gloo
def _ddp_worker(rank, world, port, out, kind):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (here, os.path.join(here, "golden"), os.path.dirname(here)):
        if p not in sys.path:
            sys.path.insert(0, p)
    import cpu_ops
    import afft_amd
    from afft_amd.optim import SGD
    afft_amd.install_as_models(patch_ddp=True)
    c, state, data, tgt, sub = _afft_case()
    with cpu_ops.installed():
        model = _afft_model(c, state, "fp32")       # eval-mode layers (the double models no dropout); the WRAPPER is in train mode
        if rank == 1:
            with torch.no_grad():
                for p in model.parameters():
                    p.add_(0.05)
        groups = _groups(model)
        clip = None
        if kind == "afft":
            opt = SGD(groups, lr=1e-2, momentum=0.9, nesterov=True, bucket_elems=8192)
        elif kind == "afft_loop_clip":      # the update waits for step(); the LOOP clips between backward() and step()
            opt = SGD(groups, lr=1e-2, momentum=0.9, nesterov=True, bucket_elems=8192, in_backward=False)
            clip = CLIP
        else:
            opt = torch.optim.SGD(groups, lr=1e-2, momentum=0.9, nesterov=True)
        ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=None, output_device=None)     # train.py:365-367
        from afft_amd.parallel import DistributedDataParallel
        assert isinstance(ddp, DistributedDataParallel) and ddp.module is model
        assert (ddp._own is None) == (kind != "torch")
        h = data[next(iter(data))].shape[0] // world
        norms = []
        _loop(ddp, opt, None, data, tgt, sub, 3, sl=slice(rank * h, (rank + 1) * h), clip=clip, norms=norms)
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    others = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(others, flat)
    assert all(torch.equal(o, others[0]) for o in others), "replicas diverged"
    if rank == 0:
        torch.save({"flat": flat, "keys": list(ddp.state_dict().keys())[:3], "norms": norms}, out)
    dist.barrier()
    dist.destroy_process_group()


CLIP = 0.05      # well below the gradient norms of the case: the clip is active on every step


@pytest.mark.parametrize("kind", ["afft", "torch", "afft_loop_clip"])
def test_two_rank_reference_loop_with_ddp_wrapper(tmp_path, kind):
    """the unchanged loop on 2 gloo ranks (half batches, rank 1 starting from other weights) == one process on the full batch:
    afft_amd.optim.SGD owns the exchange ('afft'), or the wrapper does it for torch.optim.SGD ('torch'); 'afft_loop_clip':
    afft_amd.optim.SGD(in_backward=False) with the LOOP's torch.nn.utils.clip_grad_norm_ between backward() and step()
    (train.py:254-260) -- the exchange is complete and averaged when backward() returns, so the norm the loop clips by is the
    full-batch gradient's"""
    import cpu_ops
    out = str(tmp_path / f"ddp_{kind}.pt")
    mp.spawn(_ddp_worker, args=(2, _free_port(), out, kind), nprocs=2, join=True)
    got = torch.load(out)
    assert all(k.startswith("module.") for k in got["keys"])
    c, state, data, tgt, sub = _afft_case()
    with cpu_ops.installed():
        model = _afft_model(c, state, "fp32")
        ref = torch.optim.SGD(_groups(model), lr=1e-2, momentum=0.9, nesterov=True)
        ref_norms = []
        _loop(model, ref, None, data, tgt, sub, 3, clip=CLIP if kind == "afft_loop_clip" else None, norms=ref_norms)
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    err = float((got["flat"] - flat).norm() / flat.norm())
    assert err < 1e-6, err
    if kind == "afft_loop_clip":
        assert len(ref_norms) == 3 and all(n > 2 * CLIP for n in ref_norms)          # the clip really bit
        assert all(abs(a - b) < 1e-5 * b for a, b in zip(got["norms"], ref_norms)), (got["norms"], ref_norms)
    import afft_amd
    afft_amd.set_precision("bf16")


def test_runner_lazy_metrics_are_the_references_values():
    """Runner's default ("lazy") returns, under the reference's own keys, values that wait for their device-to-host copy at first
    use: consumed the way metric_tracking.py consumes them (val * n, np.argsort, labels == l) they are the blocking Runner's values"""
    import numpy as np
    import cpu_ops
    from afft_amd.common.runner import LazyHostArray, LazyScalar, Runner
    c, state, data, tgt, sub = _afft_case()
    with cpu_ops.installed():
        model = _afft_model(c, state, "fp32")
        lazy = Runner(model, torch.device("cpu"), WTS)
        sync = Runner(model, torch.device("cpu"), WTS, async_metrics=False)
        assert lazy.async_metrics == "lazy"
        l1, m1 = lazy(_batch(data, tgt, sub), None, True)
        l2, m2 = sync(_batch(data, tgt, sub), None, True)
    assert float(l1.detach()) == float(l2.detach())
    assert set(m1) == set(m2)
    for k, v in m2.items():
        if isinstance(v, float):
            assert isinstance(m1[k], LazyScalar) and m1[k] * 3 == v * 3 and 0.0 + m1[k] == v and f"{m1[k]:.4f}" == f"{v:.4f}"
        elif isinstance(v, dict):
            assert np.array_equal(np.argsort(m1[k]["logits"], axis=1), np.argsort(v["logits"], axis=1))
            lab = m1[k]["labels"]
            assert np.array_equal(lab.reshape(-1, 1), v["labels"].reshape(-1, 1)) and np.array_equal(lab == v["labels"][0], v["labels"] == v["labels"][0])
    # on the GPU path the holders are the lazy classes; a NaN loss raises the reference's error at first use
    from afft_amd.common.runner import _Pending
    bad = LazyScalar(_Pending(torch.tensor([float("nan")])), 0)
    with pytest.raises(ValueError, match="NaN"):
        float(bad)
    arr = LazyHostArray(torch.arange(6.0).reshape(2, 3))
    assert arr.shape == (2, 3) and np.asarray(arr).sum() == 15.0


def test_a_non_finite_loss_makes_the_step_a_no_op():
    """The reference raises 'The loss is NaN!' BEFORE backward (common/runner.py:209), so a batch that produces a NaN loss never touches
    the parameters.  With lazy metrics that error surfaces later; the update kernels of the step therefore look at a device flag the
    first backward kernel writes (isfinite(loss): parallel.FusedSGD.ok / afft_sgd_fused_t.ok) and skip: parameters and momentum after a
    poisoned batch are bit-identical to before it, the optimizer keeps working afterwards, and the loss still reads NaN.  Both loop
    shapes: Trainer.step and the reference's own loop with afft_amd.optim.SGD (whose zero_grad() comes AFTER the forward pass)."""
    import cpu_ops
    from afft_amd.common.runner import Runner
    from afft_amd.optim import SGD
    from afft_amd.parallel import Trainer
    c, state, data, tgt, sub = _afft_case()
    bad = {m: d.clone() for m, d in data.items()}
    next(iter(bad.values()))[0, 0, 0] = float("nan")
    with cpu_ops.installed():
        m1 = _afft_model(c, state, "fp32")
        tr = Trainer(m1, WTS, lr=1e-2, momentum=0.9, weight_decay=1e-4, bucket_elems=8192)
        tr.step(data, {"action": tgt}, {"action": sub})
        p0, b0 = tr.flat.flat_p.clone(), tr.opt.buf.clone()
        loss, _ = tr.step(bad, {"action": tgt}, {"action": sub})
        assert float(loss) != float(loss) and float(tr.opt.ok) == 0.0
        assert torch.equal(tr.flat.flat_p, p0) and torch.equal(tr.opt.buf, b0)
        loss, _ = tr.step(data, {"action": tgt}, {"action": sub})
        assert float(loss) == float(loss) and float(tr.opt.ok) == 1.0 and not torch.equal(tr.flat.flat_p, p0)
        assert bool(torch.isfinite(tr.flat.flat_p).all())
        # the reference's loop shape
        m2 = _afft_model(c, state, "fp32")
        opt = SGD(_groups(m2), lr=1e-2, momentum=0.9, nesterov=True, bucket_elems=8192)
        runner = Runner(m2, torch.device("cpu"), WTS, compute_metrics=False)
        for batch, poisoned in ((data, False), (bad, True), (data, False)):
            before = opt.flat.flat_p.clone()
            loss, _ = runner(_batch(batch, tgt, sub), None, True)
            opt.zero_grad()
            loss.backward()
            opt.step()
            assert torch.equal(opt.flat.flat_p, before) == poisoned
        assert bool(torch.isfinite(opt.flat.flat_p).all())

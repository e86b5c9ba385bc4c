"""CPU, world_size 2 over gloo: the data-parallel gradient path (flat buffers, buckets, backward-overlapped
all-reduce launch order, learned per-bucket readiness counts) is correct by construction.  Identity checked:
the reduced gradient of two ranks with different half-batches equals the single-process gradient of the full
batch (sum-of-grads / world), on every step, with several buckets."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _toy(seed=0):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Linear(24, 40), torch.nn.Tanh(), torch.nn.Linear(40, 40), torch.nn.Tanh(),
                               torch.nn.Linear(40, 8))


def _data(step):
    g = torch.Generator().manual_seed(100 + step)
    return torch.randn(8, 24, generator=g), torch.randn(8, 8, generator=g)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from afft_amd import runtime as rt
    from afft_amd.parallel import FlatParams, GradReducer
    model = _toy()
    flat = FlatParams(model)
    red = GradReducer(flat, bucket_elems=1024)   # several buckets
    assert len(red.buckets) >= 3
    params = list(model.parameters())
    results = []
    for step in range(3):
        x, y = _data(step)
        xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
        red.begin_step()
        flat.flat_g.zero_()
        loss = torch.nn.functional.mse_loss(model(xs), ys)
        loss.backward()                       # accumulates in place into the flat views
        for p in reversed(params):            # what the wgrad epilogues do in the HIP path, in backward order
            rt.SINK.touched[id(p)] = True
            if rt.SINK.on_grad_ready is not None:
                rt.SINK.on_grad_ready(p)
        launched_early = sum(red._launched)
        red.finish_step()
        g, scale = red.grad_for_optimizer()
        results.append((g.clone() * scale, launched_early))
        # every parameter's .grad is still a view of the flat buffer
        for p, o in zip(flat.params, flat.offsets):
            assert p.grad.data_ptr() == flat.flat_g[o:].data_ptr()
    if rank == 0:
        torch.save([(r[0], r[1]) for r in results], out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    from afft_amd.parallel import FlatParams
    model = _toy()
    flat = FlatParams(model)
    for step, (g2, launched_early) in enumerate(got):
        x, y = _data(step)
        flat.flat_g.zero_()
        torch.nn.functional.mse_loss(model(x), y).backward()
        assert torch.allclose(g2, flat.flat_g, atol=1e-6), step
        # step 0 learns the per-bucket readiness counts; later steps launch buckets during backward
        assert (launched_early == 0) if step == 0 else (launched_early >= 1)


def _order_worker(rank, world, port, out, mismatch):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import random
    import time
    from afft_amd import runtime as rt
    from afft_amd.parallel import FlatParams, GradReducer
    model = _toy()
    flat = FlatParams(model)
    red = GradReducer(flat, bucket_elems=64)
    assert len(red.buckets) >= 4
    order, launch = [], red._launch

    def recording_launch(b):
        order[-1].append(b)
        launch(b)
    red._launch = recording_launch
    on_ready = red._on_ready
    rnd = random.Random(rank)

    def slow_ready(p):           # rank 1's backward lags behind rank 0's by a random 0-20 ms per gradient
        if rank == 1:
            time.sleep(rnd.random() * 0.02)
        on_ready(p)
    red._on_ready = slow_ready
    params = list(model.parameters())
    err = None
    for step in range(4):
        x, y = _data(step)
        xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
        order.append([])
        red.begin_step()
        flat.flat_g.zero_()
        torch.nn.functional.mse_loss(model(xs), ys).backward()
        ready = list(reversed(params))
        if mismatch and rank == 1 and step == 0:
            ready = ready[:-1]           # a replica whose graph misses one gradient: must be an error, not a hang
        for p in ready:
            rt.SINK.touched[id(p)] = True
            if rt.SINK.on_grad_ready is not None:
                rt.SINK.on_grad_ready(p)
        try:
            red.finish_step()
        except RuntimeError as ex:
            err = str(ex)
            break
    mine = [order, err]
    both = [None, None]
    dist.all_gather_object(both, mine)
    if rank == 0:
        torch.save(both, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mismatch", [False, True])
def test_bucket_launch_order_is_the_same_on_a_lagging_rank(tmp_path, mismatch):
    """The collectives of a step are enqueued per bucket as soon as the learned number of gradients has arrived
    (GradReducer.expected).  RCCL needs the SAME order on every rank: with rank 1's backward delayed by random sleeps the
    recorded launch order of every step is identical on both ranks (it depends on the order of the gradients, not on time),
    and buckets do launch during backward from step 2 on.  A replica that counts differently on step 1 raises on every rank
    (one all-reduce of the counts when they are learned) instead of deadlocking later.  Reference: train.py:364-368 (DDP)."""
    out = str(tmp_path / "order.pt")
    mp.spawn(_order_worker, args=(2, _free_port(), out, mismatch), nprocs=2, join=True)
    (o0, e0), (o1, e1) = torch.load(out)
    if mismatch:
        assert e0 and e1 and "disagree" in e0 and "disagree" in e1
        return
    assert e0 is None and e1 is None
    assert o0 == o1 and len(o0) == 4
    nb = len(o0[0])
    assert all(sorted(step) == list(range(nb)) for step in o0)          # every bucket exactly once per step
    assert o0[0] == list(reversed(range(nb)))                           # step 1: handed over last-to-first after backward


def test_flat_params_views_and_alignment():
    from afft_amd.parallel import FlatParams
    model = _toy(3)
    ref = [p.detach().clone() for p in model.parameters()]
    flat = FlatParams(model)
    for p, r, o in zip(model.parameters(), ref, flat.offsets):
        assert torch.equal(p.detach(), r)
        assert o % 64 == 0 and p.data_ptr() == flat.flat_p[o:].data_ptr()
    assert flat.total % 64 == 0


# ----------------------------------------------------------------------------- the REAL model + Trainer over gloo
def _afft_case():
    """t0_sa topology (SA-Fuser + GPT-2 predictor + heads + 3-term loss), labels without ignored frames so that the mean
    losses of two half-batches average to the full-batch mean (the data-parallel identity DDP relies on)."""
    from cases import CASES
    from helpers import case_tensors
    c, state, data, tgt, sub = case_tensors("t0_sa")
    B = 4
    g = torch.Generator().manual_seed(7)
    data = {m: torch.randn(B, c["T"], C, 1, 1, 1, generator=g) for m, C in c["modal_dims"].items()}
    tgt = torch.randint(0, c["num_classes"], (B,), generator=g)
    sub = torch.randint(0, c["num_classes"], (B, c["T"], 1), generator=g)
    return c, state, data, tgt, sub


def _afft_model(c, state, precision):
    import afft_amd
    from afft_amd import runtime as rt
    from afft_amd.config import make_model_cfg
    from afft_amd.models.base_model import BaseModel
    afft_amd.set_precision(precision)
    rt.set_grad_mode("sink")
    cfg = make_model_cfg(c["modal_dims"], c["d"], c["D"], fuser=c["fuser"], depth=c["depth"], num_heads=c["num_heads"],
                         fp_layers=c["fp_layers"], fp_heads=c["fp_heads"], T=c["T"], drop=0.0)
    m = BaseModel(cfg, num_classes={"action": c["num_classes"]}, class_mappings={})
    m.load_state_dict(state)
    return m.eval()


WTS = {"cls_action": 1.0, "past_cls_action": 1.0, "past_reg": 1.0}


def _afft_worker(rank, world, port, out, comm_dtype, comm_algo="allreduce"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (here, os.path.join(here, "golden"), os.path.dirname(here)):
        if p not in sys.path:
            sys.path.insert(0, p)
    import cpu_ops
    from afft_amd.parallel import Trainer
    c, state, data, tgt, sub = _afft_case()
    with cpu_ops.installed():
        model = _afft_model(c, state, "fp32")
        if rank == 1:       # a replica that starts somewhere else (unseeded init / checkpoint loaded on rank 0 only)
            with torch.no_grad():
                for p in model.parameters():
                    p.add_(0.05)
        tr = Trainer(model, WTS, lr=1e-2, momentum=0.9, weight_decay=1e-4, bucket_elems=8192, comm_dtype=comm_dtype,
                     comm_algo=comm_algo)
        assert len(tr.reducer.buckets) >= 3
        h = data[next(iter(data))].shape[0] // world
        sl = slice(rank * h, (rank + 1) * h)
        early = []
        fin = tr.reducer.finish_step

        def counting_finish():
            early.append(sum(tr.reducer._launched))     # buckets handed over DURING backward
            fin()
        tr.reducer.finish_step = counting_finish
        for _ in range(3):
            tr.step({m: d[sl] for m, d in data.items()}, {"action": tgt[sl]}, {"action": sub[sl]})
        if comm_algo == "sharded":
            assert tr.flat.sharded_layout and 0 < tr.flat.split < tr.flat.total
            nsh = sum(tr.reducer.sharded_bucket(b) for b in range(len(tr.reducer.buckets)))
            assert nsh >= 2 and nsh < len(tr.reducer.buckets), "GEMM-weight buckets are sharded, the small-parameter bucket is replicated"
            r0, r1 = tr.reducer.shard_of(*tr.reducer.buckets[0])
            mine = tr.opt.buf[r0:r1].clone()
            assert tr.reducer.masters_stale
            try:                                         # state_dict() never communicates (the reference saves on rank 0 alone,
                model.state_dict()                       # train.py:403-411): stale masters are an error, not a collective
                raise AssertionError("state_dict() on stale masters must raise")
            except RuntimeError as ex:
                assert "sync_masters" in str(ex)
            with torch.no_grad():                        # the first evaluation forward (every rank validates) brings them up to date
                model({m: d[sl] for m, d in data.items()}, mixup_fn=None, target={"action": tgt[sl]},
                      target_subclips={"action": sub[sl]}, target_subclips_ignore_index=None)
            assert not tr.reducer.masters_stale and torch.equal(tr.opt.buf[r0:r1], mine)
            sd = model.state_dict()
            # embedding tables are read as fp32 rows: they sit in the replicated region whatever their shape
            for k, p in model.named_parameters():
                if k.endswith("wpe.weight") or k.endswith("position_embeddings.weight"):
                    assert tr.flat.offsets[tr.flat.index_of()[id(p)]] >= tr.flat.split, k
    flat = tr.flat.flat_p.clone()
    buf = tr.opt.buf.clone()
    for t in (flat, buf):
        others = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(others, t)
        assert all(torch.equal(o, others[0]) for o in others), "replicas diverged"
    if rank == 0:
        byname = {k: p.detach().clone() for k, p in model.named_parameters()}
        mom = {k: tr.opt.buf[o:o + p.numel()].clone() for (k, p), o in zip(((k, p) for k, p in model.named_parameters()), [tr.flat.offsets[tr.flat.index_of()[id(p)]] for _, p in model.named_parameters()])}
        torch.save({"flat": flat, "byname": byname, "momentum": mom, "names": [k for k, _ in model.named_parameters()], "early": early}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_update_equals_the_allreduce_path_bitwise(tmp_path):
    """comm_algo = 'sharded' (reduce-scatter -> update of the rank's 1 / N slice -> all-gather, GEMM weights first in the flat
    buffers) against the all-reduce path, both on 2 gloo ranks in the exact-fp32 mode: every parameter and every momentum value
    bitwise equal by NAME after 3 steps (the two-rank sums a + b are the same bits either way), state_dict() -- which brings the
    other rank's masters and momentum up to date -- identical on both ranks."""
    res = {}
    for algo in ("allreduce", "sharded"):
        out = str(tmp_path / f"afft_{algo}.pt")
        mp.spawn(_afft_worker, args=(2, _free_port(), out, "fp32", algo), nprocs=2, join=True)
        res[algo] = torch.load(out)
    a, b = res["allreduce"], res["sharded"]
    assert a["byname"].keys() == b["byname"].keys()
    for k in a["byname"]:
        assert torch.equal(a["byname"][k], b["byname"][k]), k
        assert torch.equal(a["momentum"][k], b["momentum"][k]), k
    import afft_amd
    afft_amd.set_precision("bf16")


@pytest.mark.parametrize("comm_algo", ["allreduce", "rs_ag", "sharded"])
def test_two_rank_trainer_real_model_matches_single_process(tmp_path, comm_algo):
    """world_size 2 over gloo with the REAL BaseModel / functional sink / GradReducer / per-bucket fused SGD (the kernels
    replaced by the torch test double): rank 1 starts from different weights and must be overwritten by the construction-time
    broadcast; after 3 steps on half-batches both replicas hold bitwise-identical weights, equal to a single process
    stepping on the full batch; from step 2 the buckets are launched during backward (learned readiness counts)."""
    import cpu_ops
    from afft_amd.parallel import Trainer
    out = str(tmp_path / "afft_r0.pt")
    mp.spawn(_afft_worker, args=(2, _free_port(), out, "fp32", comm_algo), nprocs=2, join=True)
    got = torch.load(out)
    c, state, data, tgt, sub = _afft_case()
    with cpu_ops.installed():
        model = _afft_model(c, state, "fp32")
        tr = Trainer(model, WTS, lr=1e-2, momentum=0.9, weight_decay=1e-4, bucket_elems=8192)
        for _ in range(3):
            tr.step(data, {"action": tgt}, {"action": sub})
    import afft_amd
    afft_amd.set_precision("bf16")
    ref = torch.cat([p.detach().reshape(-1) for _, p in model.named_parameters()])
    have = torch.cat([got["byname"][k].reshape(-1) for k, _ in model.named_parameters()])      # by name: the sharded layout reorders the flat buffer
    err = float((have - ref).norm() / ref.norm())
    assert err < 1e-6, err
    assert got["early"][0] == 0 and all(e >= 1 for e in got["early"][1:]), got["early"]


# ----------------------------------------------------------------------------- bench.py's N > 1 report, over gloo
def _bench_report_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import argparse
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (here, os.path.join(here, "golden"), os.path.dirname(here)):
        if p not in sys.path:
            sys.path.insert(0, p)
    import bench
    import cpu_ops
    import afft_amd
    from afft_amd import config as CFG
    from afft_amd.parallel import Trainer
    CFG.BASELINE_CONFIGS["tiny"] = dict(modal_dims={"rgb": 64, "flow": 64}, common_dim=64, fp_inter_dim=64, fuser="sa", T=4)
    args = argparse.Namespace(config="tiny", comm_dtype="fp32", comm_algo="allreduce", bucket_melems=1, eval_drop=True)
    with cpu_ops.installed():
        afft_amd.set_precision("fp32")
        model, c = bench.build_model("tiny", "cpu", drop=0.0)
        model.eval()
        feats, tgt, sub = bench.make_inputs(c, 4, c["T"], rank, "cpu")
        tr = Trainer(model, WTS, comm_dtype="fp32", bucket_elems=1 << 16)
        for _ in range(2):
            tr.step(feats, tgt, sub)
        rep = bench.comm_report(args, tr, feats, tgt, sub, world, rank, torch.device("cpu"), 123.0, None, dist.barrier,
                                n_noexch=2, n_payload=4)
    if rank == 0:
        torch.save(rep, out)
    dist.barrier()
    dist.destroy_process_group()


def test_bench_comm_report_runs_on_two_ranks(tmp_path):
    """bench.py's N > 1 side measurements (exposed communication, payload-precision loss delta) cannot be run on the 1-GPU test
    box: run the same function on 2 gloo ranks with the test double, so that a typo cannot take the driver's scaling run down."""
    out = str(tmp_path / "rep.pt")
    mp.spawn(_bench_report_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    rep = torch.load(out)
    assert rep["world_size"] == 2 and rep["buckets"] >= 1
    assert rep["ms_per_step_without_exchange"] > 0 and "exposed_comm_ms" in rep
    assert set(rep["loss_after_4_steps"]) == {"fp32", "bf16"}
    assert abs(rep["loss_delta_bf16_vs_fp32_payload"]) < 0.05 * abs(rep["loss_after_4_steps"]["fp32"])


# ----------------------------------------------------------------------------- a dead / hung rank cannot hang the caller
@pytest.mark.parametrize("mode", ["die", "hang"])
def test_launcher_reports_a_dead_or_hung_rank(mode, capfd):
    """bench.launch_ranks (what `python bench.py --gpus N` runs when no launcher is around it) with two gloo ranks of which
    rank 1 dies mid-step ('die') or never enters its collective ('hang'): the call returns non-zero well inside its timeout
    budget -- rank 0's collective raises (closed connection / process-group timeout) or the launcher's own timeout ends the
    ranks -- and prints ONE JSON error line instead of hanging."""
    import json
    import time
    import bench
    here = os.path.dirname(os.path.abspath(__file__))
    t0 = time.perf_counter()
    rc = bench.launch_ranks(2, script=os.path.join(here, "scripts", "rank_dies.py"), argv=[mode], timeout=60)
    took = time.perf_counter() - t0
    out = capfd.readouterr().out
    assert rc != 0, out
    assert took < 60, took
    line = [ln for ln in out.splitlines() if ln.startswith("{") and '"error"' in ln]
    assert len(line) == 1, out
    rep = json.loads(line[0])
    assert rep["rc"] != 0 and rep["n_gpus"] == 2 and "error" in rep


# ----------------------------------------------------------------------------- N > 1: a batch that is poisoned on ONE rank is skipped by ALL
def _poison_worker(rank, world, port, out, comm_algo):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (here, os.path.join(here, "golden"), os.path.dirname(here)):
        if p not in sys.path:
            sys.path.insert(0, p)
    import cpu_ops
    from afft_amd.parallel import Trainer
    c, state, data, tgt, sub = _afft_case()
    with cpu_ops.installed():
        model = _afft_model(c, state, "fp32")
        tr = Trainer(model, WTS, lr=1e-2, momentum=0.9, weight_decay=1e-4, bucket_elems=8192, comm_dtype="fp32", comm_algo=comm_algo)
        h = data[next(iter(data))].shape[0] // world
        sl = slice(rank * h, (rank + 1) * h)
        mine = {m: d[sl].clone() for m, d in data.items()}
        bad = {m: d.clone() for m, d in mine.items()}
        if rank == 1:                                   # only rank 1's half of the batch carries the NaN
            next(iter(bad.values()))[0, 0, 0] = float("nan")
        log = []
        for feats in (mine, bad, mine):
            before = tr.flat.flat_p.clone()
            loss, _ = tr.step(feats, {"action": tgt[sl]}, {"action": sub[sl]})
            if comm_algo == "sharded":
                tr.reducer.sync_masters()
            log.append((bool(torch.equal(tr.flat.flat_p, before)), float(tr.opt.ok), float(loss) == float(loss)))
        flat = tr.flat.flat_p.clone()
        others = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(others, flat)
        assert all(torch.equal(o, others[0]) for o in others), "replicas diverged"
        assert bool(torch.isfinite(flat).all())
    torch.save(log, out + f".{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("comm_algo", ["allreduce", "sharded"])
def test_two_ranks_agree_to_skip_a_step_that_is_non_finite_on_one_of_them(tmp_path, comm_algo):
    """The device-side 'The loss is NaN!' guard (FusedSGD.ok) with a gradient exchange: rank 1's half of the second batch holds a NaN,
    rank 0's loss is finite -- the summed gradient is non-finite on both, so both must leave parameters and momentum alone
    (GradReducer._agree_ok: MIN over the ranks' flags before the first update of the step), stay bitwise equal, and train on."""
    out = str(tmp_path / "poison")
    mp.spawn(_poison_worker, args=(2, _free_port(), out, comm_algo), nprocs=2, join=True)
    logs = [torch.load(out + f".{r}") for r in range(2)]
    for r, log in enumerate(logs):
        assert [e[0] for e in log] == [False, True, False], (r, log)          # parameters unchanged exactly in the poisoned step
        assert [e[1] for e in log] == [1.0, 0.0, 1.0], (r, log)               # ... on BOTH ranks
    assert logs[0][1][2] is True and logs[1][1][2] is False                    # rank 0's own loss was finite, rank 1's was not

"""Build-container tests of the HOST logic (no GPU, no kernels): the mirrored reference modules, the autograd wiring of
afft_amd.functional (gradient sink, gradient hand-over) and the Trainer run on CPU tensors with tests/cpu_ops.py standing in
for the C-ABI wrappers, against the goldens produced by the reference.  What the kernels compute is tested on the GPU only."""
import pytest
import torch

import cpu_ops
from cases import CASES
from helpers import case_tensors, flatten_outputs, load_golden, rel_l2


def _build(c, precision):
    import afft_amd
    from afft_amd import runtime as rt
    from afft_amd.config import make_model_cfg
    from afft_amd.models.base_model import BaseModel
    afft_amd.set_precision(precision)
    rt.set_grad_mode("sink")
    cfg = make_model_cfg(c["modal_dims"], c["d"], c["D"], fuser=c["fuser"], depth=c.get("depth", 1),
                         num_heads=c["num_heads"], fp_layers=c["fp_layers"], fp_heads=c["fp_heads"],
                         fp_output_len=c.get("fp_output_len", 1), cross_attn=c.get("cross_attn", False),
                         modal_encoding=c.get("modal_encoding", False),
                         frame_level_token=c.get("frame_level_token", False), T=c["T"], cmfp=c.get("cmfp", "early"),
                         mapping=c.get("mapping", "linear"), mapping_activation=c.get("mapping_activation", "relu"),
                         mapping_layernorm=c.get("mapping_layernorm"), share_predictors=c.get("share_predictors", True),
                         share_classifiers=c.get("share_classifiers", True))
    return BaseModel(cfg, num_classes={"action": c["num_classes"]}, class_mappings={})


def _step(model, data, tgt, sub):
    from afft_amd import runtime as rt
    from afft_amd.common.runner import BasicLossAccuracy, Runner
    rt.SINK.begin_step()
    for p in model.parameters():
        p.grad = None
    out, out_t = model(data, mixup_fn=None, target={"action": tgt}, target_subclips={"action": sub},
                       target_subclips_ignore_index=None)
    losses, _ = BasicLossAccuracy(compute_metrics=False)(out, out_t["target"], out_t["target_subclips"])
    total, _ = Runner._reduce_loss(losses, {"cls_action": 1.0, "past_cls_action": 1.0, "past_reg": 1.0}, sync=False)
    total.backward()
    rt.SINK.finish_step(list(model.parameters()))
    return out, total


HOST_CASES = ["t0_sa", "t1_ca", "t2_flt", "t3_m5", "t4_cm", "t5_tsa", "t6_score", "t7_indiv", "t8_gated"]


@pytest.mark.parametrize("precision", ["fp32", "bf16", "bf16x3"])
@pytest.mark.parametrize("name", HOST_CASES)
def test_host_wiring_reproduces_reference_golden(name, precision):
    import afft_amd
    z, _ = load_golden(name)
    c, state, data, tgt, sub = case_tensors(name)
    tol = 6e-2 if precision == "bf16" else 2e-4
    with cpu_ops.installed():
        model = _build(c, precision)
        model.load_state_dict(state, strict=True)
        model.eval()
        out, total = _step(model, data, tgt, sub)
    afft_amd.set_precision("bf16")
    flat = flatten_outputs(out)
    for k in z.files:
        if k.startswith("out:") and not k.endswith("modality_attns"):
            assert rel_l2(flat[k[4:]].float(), torch.from_numpy(z[k])) < tol, k
    assert abs(float(total) - float(z["loss:total"])) < tol * max(1.0, abs(float(z["loss:total"])))
    params = dict(model.named_parameters())
    n = 0
    for k in z.files:
        if k.startswith("grad:"):
            assert rel_l2(params[k[5:]].grad, torch.from_numpy(z[k])) < (0.2 if precision == "bf16" else tol), k
            n += 1
    assert n >= 5


@pytest.mark.parametrize("name", ["t0_sa", "t1_ca", "t6_score"])
def test_fp16x2_trains_forward_fp16_two_pass_backward_bf16(name):
    """precision 'fp16x2': the forward GEMMs take fp16 hi + lo activation planes and the weights' FP16 images (two segments) --
    outputs and loss within 1e-3 of the reference golden --, the backward pass runs as the bf16 mode on bf16 copies of the saved
    activations (runtime.backward_precision): every gradient exists and sits inside the bf16 mode's bound (host wiring on the
    call-by-call path; the kernels and the composite path: GPU tests)."""
    import afft_amd
    from afft_amd import runtime as rt
    z, _ = load_golden(name)
    c, state, data, tgt, sub = case_tensors(name)
    try:
        with cpu_ops.installed():
            model = _build(c, "fp16x2")
            model.load_state_dict(state, strict=True)
            model.eval()
            out, total = _step(model, data, tgt, sub)
            assert rt.precision() == "fp16x2"          # the backward pass restored the mode it switched away from
            flat = flatten_outputs(out)
            for k in z.files:
                if k.startswith("out:") and not k.endswith("modality_attns"):
                    assert rel_l2(flat[k[4:]].float(), torch.from_numpy(z[k])) < 1e-3, k
            assert abs(float(total) - float(z["loss:total"])) < 1e-3 * max(1.0, abs(float(z["loss:total"])))
            params = dict(model.named_parameters())
            n = 0
            for k in z.files:
                if k.startswith("grad:"):
                    assert rel_l2(params[k[5:]].grad, torch.from_numpy(z[k])) < 0.2, k       # the bf16 row of the test above
                    n += 1
            assert n >= 5
    finally:
        afft_amd.set_precision("bf16")


def _grads(model):
    return {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}


def test_gradient_handover_accepted_and_off_agree():
    """bf16 mode: the LayerNorm-backward hand-over (operand + output-bias gradient of the upstream sub-layer) gives the same
    gradients as the separate cast / column-sum kernels."""
    import afft_amd
    from afft_amd import runtime as rt
    c, state, data, tgt, sub = case_tensors("t0_sa")
    res = {}
    with cpu_ops.installed():
        for on in (True, False):
            rt.set_handover(on)
            model = _build(c, "bf16")
            model.load_state_dict(state)
            model.eval()
            _step(model, data, tgt, sub)
            res[on] = _grads(model)
    rt.set_handover(True)
    afft_amd.set_precision("bf16")
    assert res[True].keys() == res[False].keys()
    for k in res[True]:
        assert rel_l2(res[True][k], res[False][k]) < 2e-2, k     # bf16 rounding of the summed copy vs the fp32 tensor


def test_gradient_handover_rejected_does_not_double_count():
    """A sub-layer output with TWO consumers: autograd sums their gradients into a new tensor, so the upstream sub-layer must
    turn the hand-over down -- and its output-bias gradient must then be the column sum of the REAL dy, once (the hand-over
    used to commit colsum(dx) to the sink at emission time and the fall-back added colsum(dy) on top)."""
    import afft_amd
    from afft_amd import functional as F_
    from afft_amd import runtime as rt
    torch.manual_seed(0)
    R, d, L_, H = 12, 64, 4, 2
    mk = lambda *s: torch.nn.Parameter(torch.randn(*s) * 0.1)   # noqa: E731
    P = dict(l1w=mk(d), l1b=mk(d), wq=mk(3 * d, d), wp=mk(d, d), bp=mk(d), l2w=mk(d), l2b=mk(d), w1=mk(4 * d, d), b1=mk(4 * d),
             w2=mk(d, 4 * d), b2=mk(d))
    x = torch.randn(R, d)
    out = {}
    with cpu_ops.installed():
        afft_amd.set_precision("bf16")
        rt.set_grad_mode("sink")
        for on in (True, False):
            rt.set_handover(on)
            rt.SINK.begin_step()
            for p in P.values():
                p.grad = torch.full_like(p, 7.0)    # stale values: first touch must overwrite
            xin = x.clone().requires_grad_(True)
            y, _ = F_.AttnSublayer.apply(xin, P["l1w"], P["l1b"], P["wq"], None, P["wp"], P["bp"], L_, H, "none", 1e-6, False)
            z = F_.MLPSublayer.apply(y, P["l2w"], P["l2b"], P["w1"], P["b1"], P["w2"], P["b2"], 1e-6, "erf", False)
            (z.sum() + (y * y).sum()).backward()       # second consumer of y
            out[on] = {k: p.grad.clone() for k, p in P.items()}
    rt.set_handover(True)
    for k in out[True]:
        assert rel_l2(out[True][k], out[False][k]) < 2e-2, k
    assert float(out[True]["bp"].abs().max()) < 1e3


def test_trainer_on_cpu_matches_torch_sgd():
    """Trainer (flat buffers + sink + per-bucket fused SGD inside backward) against torch.optim.SGD on autograd-mode
    gradients of the same model, two steps."""
    import afft_amd
    from afft_amd import runtime as rt
    from afft_amd.parallel import Trainer
    c, state, data, tgt, sub = case_tensors("t0_sa")
    wts = {"cls_action": 1.0, "past_cls_action": 1.0, "past_reg": 1.0}
    with cpu_ops.installed():
        m1 = _build(c, "fp32")
        m1.load_state_dict(state)
        m1.eval()
        tr = Trainer(m1, wts, lr=1e-2, momentum=0.9, weight_decay=1e-4, bucket_elems=4096, overlap_optimizer=True)
        for _ in range(2):
            tr.step(data, {"action": tgt}, {"action": sub})
        m2 = _build(c, "fp32")
        m2.load_state_dict(state)
        m2.eval()
        rt.set_grad_mode("autograd")
        opt = torch.optim.SGD(m2.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-4, nesterov=True)
        from afft_amd.common.runner import BasicLossAccuracy, Runner
        for _ in range(2):
            opt.zero_grad()
            o, ot = m2(data, mixup_fn=None, target={"action": tgt}, target_subclips={"action": sub},
                       target_subclips_ignore_index=None)
            ls, _ = BasicLossAccuracy(False)(o, ot["target"], ot["target_subclips"])
            tot, _ = Runner._reduce_loss(ls, wts, sync=False)
            tot.backward()
            opt.step()
        rt.set_grad_mode("sink")
    afft_amd.set_precision("bf16")
    p2 = dict(m2.named_parameters())
    for k, p in m1.named_parameters():
        assert rel_l2(p, p2[k]) < 1e-5, k


def test_multi_crop_7d_input_averages_crops():
    """BaseModel.forward with 7-D (B, #clips, #crops, C, 1, 1, 1) inputs (models/base_model.py:68-119): a single crop equals
    the 6-D call; with 3 crops every output is the mean of the per-crop outputs, the attention maps are the first crop's, a
    modality delivered with one crop is shared by all crops, and the targets are handed through."""
    import afft_amd
    c, state, data, tgt, sub = case_tensors("t0_sa")
    g = torch.Generator().manual_seed(5)
    kw = dict(mixup_fn=None, target={"action": tgt}, target_subclips={"action": sub}, target_subclips_ignore_index=None)
    with cpu_ops.installed(), torch.no_grad():
        model = _build(c, "fp32")
        model.load_state_dict(state)
        model.eval()
        base, t0 = model(data, **kw)
        one, _ = model({m: d.unsqueeze(2) for m, d in data.items()}, **kw)
        crops = {m: torch.stack([d, d + 0.1 * torch.randn(d.shape, generator=g), d * 0.5], dim=2) for m, d in data.items()}
        crops["flow"] = data["flow"].unsqueeze(2)          # one crop only: cycled
        multi, t3 = model(crops, **kw)
        per = [model({m: (x[:, :, i] if x.shape[2] > 1 else x[:, :, 0]) for m, x in crops.items()}, **kw)[0] for i in range(3)]
    afft_amd.set_precision("bf16")
    assert t3["target"]["action"] is tgt and t0["target_subclips"]["action"] is sub
    for key, by_mod in base.items():
        for m, v in by_mod.items():
            if key == "attentions":
                assert torch.equal(multi[key][m]["modality_attns"], per[0][key][m]["modality_attns"])
                continue
            assert torch.equal(one[key][m], v), key
            want = torch.stack([p[key][m] for p in per]).mean(0)
            assert torch.allclose(multi[key][m], want, atol=1e-6), key
    with pytest.raises(NotImplementedError):
        model({m: d[:, :, :, 0] for m, d in data.items()}, **kw)


def test_two_threads_forward_concurrently_without_sharing_state():
    """The reference's eval path is nn.DataParallel (test.py:130): one Python thread per replica calls forward at the same time.
    Everything afft_amd.functional remembers between calls (noted sub-layer output for the hand-over, side-stream block,
    pending notifications) is per thread: two threads running different models interleaved get the results of running them
    one after the other."""
    import threading
    import afft_amd
    cases = ["t0_sa", "t1_ca"]
    built = {}
    with cpu_ops.installed():
        for name in cases:
            c, state, data, tgt, sub = case_tensors(name)
            m = _build(c, "bf16")
            m.load_state_dict(state)
            built[name] = (m.eval(), data, tgt, sub)
        kw = lambda tgt, sub: dict(mixup_fn=None, target={"action": tgt}, target_subclips={"action": sub},   # noqa: E731
                                   target_subclips_ignore_index=None)
        with torch.no_grad():
            serial = {n: flatten_outputs(m(d, **kw(t, s))[0]) for n, (m, d, t, s) in built.items()}
        got, errs = {}, []
        gate = threading.Barrier(2)

        def run(name):
            try:
                m, d, t, s = built[name]
                gate.wait()
                with torch.no_grad():
                    for _ in range(3):
                        got[name] = flatten_outputs(m(d, **kw(t, s))[0])
            except Exception as e:      # noqa: BLE001
                errs.append(e)
        ths = [threading.Thread(target=run, args=(n,)) for n in cases]
        [t.start() for t in ths]
        [t.join() for t in ths]
    afft_amd.set_precision("bf16")
    assert not errs, errs
    for n in cases:
        for k, v in serial[n].items():
            assert torch.equal(got[n][k], v), (n, k)


def test_weight_image_registry_does_not_grow_with_steps():
    """runtime._wlist holds ONE weak reference per parameter: a bf16x3 split that is dropped after every optimizer step
    (FusedSGD.end_step -> invalidate_weight_images) and rebuilt by the next forward must not append an entry per step
    (the list - and the cost of invalidate - grew linearly with the steps taken)."""
    from afft_amd import ops, runtime as rt

    class FakeSplit:
        def __init__(self, x, f16=False):
            self.planes, self.rows, self.cols, self.f16 = x.clone(), x.shape[0], x.shape[1], f16

    saved, ops.Split = ops.Split, FakeSplit
    try:
        import gc
        gc.collect()
        rt.invalidate_weight_images()       # drops the entries of parameters earlier tests left behind
        ws = [torch.nn.Parameter(torch.randn(8, 8)) for _ in range(3)]
        before = len(rt._wlist)
        for _ in range(6):
            for w in ws:
                sp = rt.weight_split(w)
                assert rt.weight_split(w) is sp          # cached within a step
            rt.invalidate_weight_images()
            assert all(w._afft_split is None for w in ws)
        assert len(rt._wlist) - before == 3
        del ws, w, sp
        gc.collect()
        rt.invalidate_weight_images()
        assert len(rt._wlist) == before
    finally:
        ops.Split = saved


def test_zero_weighted_loss_terms_leave_the_total_and_the_graph():
    """a loss weight <= 0 (expts/05: past_cls_action=0) DROPS the term as in the reference (runner.py:205-207): a NaN in
    it does not poison the total, nothing flows back through its branch, its mean is still logged; all weights <= 0 is an error."""
    from afft_amd.common.runner import Runner
    with cpu_ops.installed():
        a = torch.tensor([1.0, 3.0], requires_grad=True)
        b = torch.tensor([float("nan"), 2.0], requires_grad=True)
        total, parts = Runner._reduce_loss({"cls_action": a, "past_cls_action": b}, {"cls_action": 2.0, "past_cls_action": 0.0}, sync=False)
        assert float(total) == 4.0 and float(parts["cls_action"]) == 2.0 and parts["past_cls_action"] != parts["past_cls_action"]
        total.backward()
        assert torch.equal(a.grad, torch.tensor([1.0, 1.0])) and b.grad is None
        with pytest.raises(RuntimeError, match="every loss weight"):
            Runner._reduce_loss({"cls_action": a.detach()}, {"cls_action": 0.0}, sync=False)


def test_interface_edges_arbitrary_mask_mem_dim_qkv_bias_reproduce_the_reference():
    """What the AFFT configurations never use but the reference's classes accept (missing #4): Block with an ARBITRARY additive
    (N, N) mask (models/transformerblock.py:26-28) and DecoderBlock(mem_dim != dim, qkv_bias=True) (:41-50) -- the mirrored modules
    route them call by call (attention core with an additive table, biased q / k / v projections, a memory of another width); outputs,
    attention maps and every gradient against the reference's own (tests/golden/e0_edges.npz), host wiring on the torch test double."""
    import afft_amd
    from afft_amd import runtime as rt
    from helpers import edge_error, edge_fixture, run_edge_modules
    z, meta, states, inputs = edge_fixture()
    afft_amd.set_precision("fp32")
    rt.set_grad_mode("sink")
    with cpu_ops.installed():
        got = run_edge_modules(torch.device("cpu"), states, inputs, meta)
    afft_amd.set_precision("bf16")
    assert set(got) == {k for k in z.files if k not in ("shapes", "meta", "block.mask", "dec.mask")}
    for k, t in got.items():
        assert t is not None, k
        assert edge_error(t, torch.from_numpy(z[k])) < 2e-5, (k, edge_error(t, torch.from_numpy(z[k])))


def test_two_losses_on_one_half_of_the_merged_logits_add_up():
    """SplitRows hands each half of the merged classifier output ONE landing slice for its loss gradient.  Two cross-entropy
    terms on the SAME half (two targets on one tensor) must not both write it in place -- autograd would add two aliasing views and
    return twice the last gradient.  The second writer of a backward pass gets its own tensor; the sum equals autograd's."""
    import afft_amd
    from afft_amd import functional as F_, runtime as rt
    from afft_amd.common.runner import MultiDimCrossEntropy
    afft_amd.set_precision("fp32")
    rt.set_grad_mode("sink")
    torch.manual_seed(5)
    B, T, C = 3, 4, 7
    x0 = torch.randn(B, T + 1, C)
    t1 = torch.randint(0, C, (B, T))
    t2 = torch.randint(0, C, (B, T))
    t3 = torch.randint(0, C, (B, 1))
    ce = MultiDimCrossEntropy()
    with cpu_ops.installed():
        for rep in range(2):      # twice through one graph shape: the per-pass 'written' flags are reset by SplitRows.backward
            x = x0.clone().requires_grad_(True)
            past, fut = F_.split_rows(x * 1.0, T)
            loss = ce(past, t1).mean() + 0.5 * ce(past, t2).mean() + ce(fut, t3).mean()
            loss.backward()
            xr = x0.clone().requires_grad_(True)
            lr = (torch.nn.functional.cross_entropy(xr[:, :T].reshape(-1, C), t1.reshape(-1))
                  + 0.5 * torch.nn.functional.cross_entropy(xr[:, :T].reshape(-1, C), t2.reshape(-1))
                  + torch.nn.functional.cross_entropy(xr[:, T:].reshape(-1, C), t3.reshape(-1)))
            lr.backward()
            assert abs(float(loss) - float(lr)) < 1e-5
            assert rel_l2(x.grad, xr.grad) < 1e-5, rel_l2(x.grad, xr.grad)
    afft_amd.set_precision("bf16")


def test_one_pass_sites_parse_default_and_width_gate():
    """runtime.one_pass_sites: the default set (the predictor's four GEMM sites and the fusers' fc2), group names, the flags a composite sub-layer
    receives (AFFT_F16X2_ONE_PASS_1 = 4, _2 = 8, _ATTN = 16) and the width gate below which every site keeps its second pass."""
    from afft_amd import runtime as rt
    saved = rt.one_pass_sites()
    try:
        rt.set_one_pass_sites("conv1d.qkv,conv1d.proj,conv1d.fc1,conv1d.fc2,linear.fc2")
        assert rt.one_pass_flags(True, 2048, "qkv", "proj", "attn") == 12 and rt.one_pass_flags(True, 2048, "fc1", "fc2") == 12
        assert rt.one_pass_flags(False, 2048, "qkv", "proj", "attn") == 0 and rt.one_pass_flags(False, 2048, "fc1", "fc2") == 8
        assert rt.one_pass_flags(False, 128, "fc1", "fc2") == 0 and rt.one_pass_flags(True, 512, "qkv", "proj", "attn") == 0
        rt.set_one_pass_sites("linear")
        assert rt.one_pass_sites() == {"linear.qkv", "linear.attn", "linear.proj", "linear.fc1", "linear.fc2"}
        assert rt.one_pass_flags(False, 1024, "qkv", "proj", "attn") == 28
        rt.set_one_pass_sites(["attn", "conv1d.fc2"])
        assert rt.one_pass_sites() == {"linear.attn", "conv1d.attn", "conv1d.fc2"}
        rt.set_one_pass_sites("")
        assert rt.one_pass_sites() == frozenset() and rt.one_pass_flags(True, 4096, "qkv", "proj", "attn") == 0
        with pytest.raises(ValueError, match="unknown one-pass site"):
            rt.set_one_pass_sites("linear.fc3")
    finally:
        rt.set_one_pass_sites(saved)

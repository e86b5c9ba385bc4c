"""CPU: the oracle reproduces every golden vector produced by the reference itself
(tests/golden/make_golden.py).  This is what pins the oracle (the reference has no tests)."""
import pytest
import torch

from cases import CASES, FULL_CASES, oracle_cfg
from helpers import (case_tensors, compact_error, flatten_outputs, full_case_tensors, full_gradient_errors, load_golden, rel_l2,
                     surrogate)
from oracle import afft_oracle as O

TOL = 2e-5  # fp32 reference vs fp32 restatement; measured worst 1.4e-6


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_matches_reference_golden(name):
    z, _ = load_golden(name)
    c, state, data, tgt, sub = case_tensors(name)
    P = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    cfg = oracle_cfg(c)
    if c.get("soft"):
        f3 = {m: torch.flatten(d.mean([-1, -2]).permute(0, 1, 3, 2), 1, 2) for m, d in data.items()}
        f3, t_s, s_s, ign = O.mixup(f3, tgt, sub, c["num_classes"], c["label_smoothing"], c["lam"])
        out = O.cmfp_early(P, f3, cfg)
        total, losses = O.loss(out, t_s, s_s, soft=True, ignore=ign)
    elif c.get("fp_output_len", 1) > 1:
        out = O.base_model_forward(P, data, cfg)
        total, losses = surrogate(out), {}
    else:
        out = O.base_model_forward(P, data, cfg)
        total, losses = O.loss(out, tgt, sub)
    flat = flatten_outputs(out)
    n_checked = 0
    for k in z.files:
        if k.startswith("out:"):
            key = k[4:]
            if key == "attentions/modality_attns" and c["fuser"] == "ca":
                continue
            ref = torch.from_numpy(z[k])
            assert flat[key].shape == ref.shape, (key, flat[key].shape, ref.shape)
            assert rel_l2(flat[key], ref) < TOL, key
            n_checked += 1
    assert n_checked >= 6
    assert abs(float(total) - float(z["loss:total"])) < TOL * max(1.0, abs(float(z["loss:total"])))
    for k, v in losses.items():
        assert abs(float(v) - float(z["loss:" + k])) < TOL * max(1.0, abs(float(z["loss:" + k]))), k
    total.backward()
    ng = 0
    for k in z.files:
        if k.startswith("grad:"):
            g = P[k[5:]].grad
            assert g is not None, k
            assert rel_l2(g, torch.from_numpy(z[k])) < 5e-5, k
            ng += 1
    assert ng >= 5
    # norms of ALL parameter gradients
    names = [str(s) for s in z["gradnames"]]
    for nm, gn in zip(names, z["gradnorm"]):
        assert abs(float(P[nm].grad.norm()) - gn) < 1e-4 * max(gn, 1e-3), nm


@pytest.mark.parametrize("name", list(FULL_CASES))
def test_oracle_matches_reference_at_full_size(name):
    """The reference itself at the real widths (cases.FULL_CASES: BASELINE cfg1, the EK100 widths of expts/01, cfg2 = the bench
    workload, cfg4 = CA-Fuser, cfg5 = five modalities at T = 32; 388-614 M parameters, head dims 256 / 512, 3806 classes, 6 + 6 layers): every output tensor, the
    three losses and the gradient of EVERY parameter (norm + a 256-element strided sample) of the oracle against the fixture."""
    c, z, state, data, tgt, sub = full_case_tensors(name)
    P = {k: v.requires_grad_(True) for k, v in state.items()}
    out = O.base_model_forward(P, data, oracle_cfg(c))
    total, losses = O.loss(out, tgt, sub)
    flat = flatten_outputs(out)
    keys = sorted({k.split(":")[1] for k in z.files if k.startswith("out:")})
    assert len(keys) >= 6
    for key in keys:
        assert compact_error(flat[key], z, "out:" + key) < TOL, key
    assert abs(float(total) - float(z["loss:total"])) < TOL * max(1.0, abs(float(z["loss:total"])))
    for k, v in losses.items():
        assert abs(float(v) - float(z["loss:" + k])) < TOL * max(1.0, abs(float(z["loss:" + k]))), k
    total.backward()
    errs = full_gradient_errors({k: p.grad for k, p in P.items() if p.grad is not None}, z)
    assert len(errs) >= 140
    worst = max(errs, key=errs.get)
    assert errs[worst] < 1e-4, (worst, errs[worst])


def test_kat_activations_and_softmax():
    """Known-answer checks for the constants that differ between the two transformers."""
    x = torch.tensor([-3.0, -1.0, 0.0, 0.5, 2.0])
    assert torch.allclose(O.gelu_erf(x), torch.nn.functional.gelu(x), atol=1e-6)
    assert torch.allclose(O.gelu_tanh(x), torch.nn.functional.gelu(x, approximate="tanh"), atol=1e-6)
    assert abs(float(O.gelu_erf(torch.tensor(1.0))) - 0.8413447) < 1e-6
    assert abs(float(O.gelu_tanh(torch.tensor(1.0))) - 0.8411920) < 1e-6
    m = O.make_mask("causal", 3)
    p = torch.softmax(torch.zeros(3, 3) + m, -1)
    assert torch.equal(p[0], torch.tensor([1.0, 0.0, 0.0]))  # masked probabilities are exactly 0
    d = O.make_mask("diag", 3)
    assert torch.softmax(torch.zeros(3, 3) + d, -1)[1, 1] == 0
    y5 = O.layer_norm(torch.tensor([[1.0, 2.0, 4.0]]), None, None, 1e-5)
    y6 = O.layer_norm(torch.tensor([[1.0, 2.0, 4.0]]), None, None, 1e-6)
    assert not torch.equal(y5, y6) and torch.allclose(y5, y6, atol=1e-5)
    oh = O.one_hot(torch.tensor([2]), 4, 0.4)
    assert torch.allclose(oh, torch.tensor([[0.1, 0.1, 0.7, 0.1]]))


def _eval_fixture():
    import os
    import numpy as np
    from closed_form import eval_inputs
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "m0_marginalize.npz"))
    return z, eval_inputs()


def test_eval_path_oracle_and_host_metrics_match_reference_golden():
    """m0_marginalize.npz = the reference's challenge.marginalize_verb_noun on a stub dataset: the oracle restatement and the
    product's host-side accuracy bookkeeping (afft_amd.challenge: numpy, no GPU involved) reproduce it."""
    import numpy as np
    import pandas as pd
    from afft_amd import challenge as CH
    from oracle import afft_oracle as O
    z, (logits, mv, mn, a_lab, v_lab, n_lab) = _eval_fixture()
    acc, scores = O.marginalize_verb_noun(logits, mv, mn, v_lab, n_lab, a_lab)
    for got, key in zip(scores, ("verb", "noun", "action")):
        assert np.allclose(got, z[key], rtol=1e-6, atol=1e-7), key
    want = dict(zip([str(k) for k in z["acc_names"]], z["acc_values"]))
    for k, v in want.items():
        assert (np.isnan(v) and np.isnan(acc[k])) or abs(acc[k] - v) < 1e-9, k

    class _DS:
        df = pd.DataFrame(dict(verb_class=v_lab, noun_class=n_lab, action_class=a_lab))
        classes_manyshot = {}
    got = CH.compute_accuracies_epic([z["verb"], z["noun"], z["action"]], _DS)
    assert set(got) == set(want)
    for k, v in want.items():
        assert (np.isnan(v) and np.isnan(got[k])) or abs(got[k] - v) < 1e-9, (k, got[k], v)


def test_unseen_tail_and_manyshot_recalls_match_reference_golden(tmp_path):
    """m1_unseen_tail.npz = the reference's compute_accuracies_epic(..., compute_manyshot_unseen_tail=True) on an EPIC-100 stub
    (challenge.py:109-193) with closed-form narration ids, RULSTM id tables and many-shot subsets: afft_amd.challenge reproduces all
    18 numbers (top-1 / top-5 / recall, many-shot, tail, unseen participants) from the same tables on disk."""
    import os
    import numpy as np
    import pandas as pd
    from afft_amd import challenge as CH
    from closed_form import unseen_tail_tables
    z, (logits, mv, mn, a_lab, v_lab, n_lab) = _eval_fixture()
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "m1_unseen_tail.npz"))
    want = dict(zip([str(k) for k in g["acc_names"]], g["acc_values"]))
    ids, tables, manyshot = unseen_tail_tables(len(a_lab), logits.shape[1])
    for fn, rows in tables.items():
        (tmp_path / fn).write_text("\n".join(rows) + "\n")

    class _DS:
        df = pd.DataFrame(dict(verb_class=v_lab, noun_class=n_lab, action_class=a_lab, narration_id=ids))
        classes_manyshot = manyshot
        version = CH.EPIC100_VERSION
        rulstm_annotation_dir = str(tmp_path)
    got = CH.compute_accuracies_epic([z["verb"], z["noun"], z["action"]], _DS, compute_manyshot_unseen_tail=True)
    assert set(got) == set(want) and len(want) == 18
    for k, v in want.items():
        assert abs(got[k] - v) < 1e-9, (k, got[k], v)
    _DS.version = 0.1                       # EPIC-55: many-shot only, no unseen / tail keys (challenge.py:190)
    assert not any("tail" in k or "unseen" in k for k in CH.compute_accuracies_epic([z["verb"], z["noun"], z["action"]], _DS, True))


def test_oracle_reproduces_the_reference_on_the_interface_edges():
    """tests/golden/e0_edges.npz (the reference's Block with an arbitrary additive mask; DecoderBlock(mem_dim != dim, qkv_bias=True)):
    the oracle's block / decoder_block from closed-form weights and inputs, outputs + input gradients."""
    import torch
    from helpers import edge_error as rel_l2, edge_fixture
    from oracle import afft_oracle as O
    z, meta, states, inputs = edge_fixture()
    H = meta["heads"]
    P = {("b." + k): v.clone().requires_grad_(True) for k, v in states["block"].items()}
    x = inputs["block.x"].clone().requires_grad_(True)
    y, attn = O.block(P, "b.", x, H, inputs["block.mask"])
    y.pow(2).mean().backward()
    for k, t in (("block.y", y), ("block.attn", attn), ("block.dx", x.grad)):
        assert rel_l2(t, torch.from_numpy(z[k])) < 2e-5, k
    for k, p in P.items():
        assert rel_l2(p.grad, torch.from_numpy(z["block.grad." + k[2:]])) < 2e-5, k
    P2 = {("d." + k): v.clone().requires_grad_(True) for k, v in states["dec"].items()}
    x2, mem = inputs["dec.x"].clone().requires_grad_(True), inputs["dec.mem"].clone().requires_grad_(True)
    y2 = O.decoder_block(P2, "d.", x2, mem, H, inputs["dec.mask"])
    y2.pow(2).mean().backward()
    for k, t in (("dec.y", y2), ("dec.dx", x2.grad), ("dec.dmem", mem.grad)):
        assert rel_l2(t, torch.from_numpy(z[k])) < 2e-5, k
    for k, p in P2.items():
        assert rel_l2(p.grad, torch.from_numpy(z["dec.grad." + k[2:]])) < 2e-5, k

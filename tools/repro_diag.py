"""Where two runs of the same three training steps differ (tests/test_model_gpu.py: full-size reproducibility): per parameter,
how many elements of the fp32 masters differ between runs.  usage: python tools/repro_diag.py [cfg2|ek100] [fused 0/1] [steps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import afft_amd  # noqa: E402
from afft_amd import dropout as D_, runtime as rt  # noqa: E402
from afft_amd.config import BASELINE_CONFIGS, make_model_cfg  # noqa: E402
from afft_amd.models.base_model import BaseModel  # noqa: E402
from afft_amd.parallel import Trainer  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
fused = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
c = BASELINE_CONFIGS[name]
B, T, K = 64, c["T"], 3806
afft_amd.set_precision("bf16")
rt.set_grad_mode("sink")
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(8)
feats = {m: torch.randn(B, T, C, 1, 1, 1, generator=g).to(dev) for m, C in c["modal_dims"].items()}
tgt = {"action": torch.randint(0, K, (B,), generator=g).to(dev)}
sub = {"action": torch.randint(0, K, (B, T, 1), generator=g).to(dev)}
if os.environ.get("WS0") == "1":      # no split-K anywhere: the scratch is too small for any split problem
    from afft_amd import ops
    ops.set_workspace_bytes(1 << 20)
wts = {"cls_action": 1.0, "past_cls_action": 1.0, "past_reg": 1.0}


PRE = []
TAPS = []      # per run: list of (tag, {name: clone}) for the fuser's composite backward calls (R = 5120)


def _tap(kind, t):
    if t["R"] != 5120 or os.environ.get("TAP") != "1":
        return
    R, d = t["R"], t["d"]
    if kind == "attn_bwd_pre":
        from afft_amd import runtime as rt_
        PRE[:] = [rt_.weight_images(t["w_proj"]).clone(), rt_.weight_images(t["w_qkv"]).clone()]
        return
    sc = t["scratch"]
    if kind == "attn_bwd":
        parts = {"dya": sc[:R * d], "dao": sc[R * d:2 * R * d], "dqkv": sc[2 * R * d:5 * R * d], "dxn": sc[5 * R * d:6 * R * d]}
        shapes = {"dya": d, "dao": d, "dqkv": 3 * d, "dxn": d}
    else:
        h = t["hidden"]
        parts = {"dya": sc[:R * d], "dxn": sc[R * d:2 * R * d], "du": sc[2 * R * d:2 * R * d + R * h]}
        shapes = {"dya": d, "dxn": d, "du": h}
    if t["shadow"] is not None:       # the cast slot is unused then: the sub-layer reads the shadow the LayerNorm backward emitted
        parts["dya"] = t["shadow"].act.buf[:R * d]
    rec = {k: v.view(R, shapes[k]).clone() for k, v in parts.items()}
    rec["dx"] = t["dx"].clone()
    rec["dy"] = t["dy"].clone()
    rec["saved"] = t["saved"].clone().view(R, -1)
    rec["x"] = t["x"].clone()
    rec["partial"] = t["partial"].clone().view(-1, d)
    if kind == "attn_bwd":
        rec["w_proj_img"], rec["w_qkv_img"] = PRE
    TAPS[-1].append((kind, rec))


from afft_amd import functional as F_  # noqa: E402
F_._TAP = _tap


def run():
    TAPS.append([])
    rt.set_fused_sgd(fused)
    D_.manual_seed(17)
    torch.manual_seed(9)
    cfg = make_model_cfg(c["modal_dims"], c["common_dim"], c["fp_inter_dim"], fuser=c["fuser"], T=T)
    model = BaseModel(cfg, {"action": K}, {}).to(dev).train()
    tr = Trainer(model, wts, lr=1e-2, bucket_elems=int(os.environ.get("BUCKET", str(32 * 1024 * 1024))))
    if os.environ.get("DEFER") == "1":
        tr.reducer.defer_all = True
    snaps = []
    can = []
    if os.environ.get("CANARY") == "1":      # zero-filled blocks between the transient allocations: a stray write shows up as a non-zero
        can = [torch.zeros(16 << 20, dtype=torch.float32, device=dev) for _ in range(96)]
        del can[::2]
    for _ in range(steps):
        loss = tr.step(feats, tgt, sub)
        if os.environ.get("NOSYNC") != "1":
            torch.cuda.synchronize()
        snaps.append((tr.flat.flat_p.clone(), tr.flat.flat_g.clone(), float(loss[0] if isinstance(loss, tuple) else loss)))
    torch.cuda.synchronize()
    for i, cn in enumerate(can):
        nz = cn.nonzero().flatten()
        if nz.numel():
            print(f"CANARY {i} (at {cn.data_ptr():#x}): {nz.numel()} non-zero words, first at {int(nz[0])}, last at {int(nz[-1])}, "
                  f"values {cn[nz[:4]].tolist()}, index steps {(nz[1:9] - nz[:8]).tolist()}", flush=True)
    del can
    names = {id(p): n for n, p in model.named_parameters()}
    layout = [(names[id(p)], o, p.numel()) for p, o in zip(tr.flat.params, tr.flat.offsets)]
    del tr, model
    torch.cuda.empty_cache()
    return snaps, layout


NRUNS = int(os.environ.get("RUNS", "3"))


def compare(first, other, tag):
    layout = first[1]
    for s in range(steps):
        if all(torch.equal(first[0][s][k], other[0][s][k]) for k in (0, 1)):
            continue
        for what, k in (("grad", 1), ("param", 0)):
            a, b = first[0][s][k], other[0][s][k]
            bad = []
            for n, o, sz in layout:
                d = (a[o:o + sz] != b[o:o + sz]).sum().item()
                if d:
                    bad.append((n, d, sz, (a[o:o + sz] - b[o:o + sz]).abs().max().item()))
            print(f"run0 vs run{tag} step {s} {what}: {len(bad)} parameters differ (loss {first[0][s][2]!r} vs {other[0][s][2]!r})")
            for t in bad[-14:]:
                print("   ", t)
        return
    print(f"run0 vs run{tag}: identical over {steps} steps", flush=True)


def compare_taps(a, b, tag):
    """first tapped buffer (backward order) that differs between two runs, and where"""
    for j, ((ka, ra), (kb, rb)) in enumerate(zip(a, b)):
        for name in ("w_proj_img", "w_qkv_img", "x", "saved", "dy", "dya", "dao", "du", "dqkv", "dxn", "partial", "dx"):
            if name not in ra:
                continue
            if not torch.equal(ra[name], rb[name]):
                dm = ra[name] != rb[name]
                rows, cols = dm.any(1).nonzero().flatten(), dm.any(0).nonzero().flatten()
                cb = sorted(set((cols // 64).tolist()))
                rbk = sorted(set((rows // 256).tolist()))
                print(f"  TAP run{tag}: call {j} ({ka}, {j % 12}th of its step) buffer {name}: {int(dm.sum())} elements differ, rows {int(rows.min())}..{int(rows.max())} "
                      f"({len(rows)} rows, 256-row blocks {rbk[:12]}), 64-col blocks {cb[:24]}, max abs diff "
                      f"{float((ra[name].float() - rb[name].float()).abs().max()):.3e}, ref max {float(ra[name].float().abs().max()):.3e}", flush=True)
                if name == "dao":      # which run holds what the GEMM gives on the tapped inputs?
                    from afft_amd import ops
                    torch.cuda.synchronize()
                    for who, r_ in (("run0", ra), (f"run{tag}", rb)):
                        ref = torch.empty_like(r_["dao"])
                        ops.gemm(r_["dya"], r_["w_proj_img"], ref)
                        torch.cuda.synchronize()
                        print(f"     {who}: tapped dao {'==' if torch.equal(ref, r_['dao']) else '!='} dya @ W_proj image recomputed alone "
                              f"({int((ref != r_['dao']).sum())} elements differ)", flush=True)
                return
    print(f"  TAP run{tag}: all tapped buffers identical", flush=True)


first = run()
first_taps = TAPS[0]
for i in range(1, NRUNS):
    junk = None
    if os.environ.get("CHURN") == "1":      # vary what the caching allocator hands out from run to run
        gg = torch.Generator().manual_seed(i)
        junk = [torch.empty(int(torch.randint(1, 64, (1,), generator=gg)) << 20, device=dev) for _ in range(16)]
        del junk[::2]
    other = run()
    compare(first, other, i)
    if os.environ.get("TAP") == "1":
        if not all(torch.equal(first[0][s_][1], other[0][s_][1]) for s_ in range(steps)):
            compare_taps(first_taps, TAPS[-1], i)
        if i == 1:
            second, second_taps = other, TAPS[-1]
        else:
            print(f"  (run1 vs run{i}: grads {'identical' if all(torch.equal(second[0][s_][1], other[0][s_][1]) for s_ in range(steps)) else 'differ'})")
            TAPS[-1] = None
    del junk, other

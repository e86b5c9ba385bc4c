"""One tiny forward + backward of the hot path on cuda:0 (SA-Fuser + GPT-2 predictor + heads + loss),
checked against the CPU oracle.  Used by __graft_entry__.smoke()."""
from __future__ import annotations

import os
import sys


def run_smoke():
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests"), os.path.join(root, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import afft_amd
    from afft_amd import runtime as rt
    from afft_amd.common.runner import BasicLossAccuracy, Runner
    from afft_amd.config import make_model_cfg
    from afft_amd.models.base_model import BaseModel
    from oracle import afft_oracle as O  # checker only

    if not torch.cuda.is_available():
        raise RuntimeError("smoke() needs a GPU: the HIP path has no CPU fallback")
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    mods = {"rgb": 128, "objects": 40, "audio": 128, "flow": 128}
    B, T, K = 4, 8, 37
    wts = {"cls_action": 1.0, "past_cls_action": 1.0, "past_reg": 1.0}
    ocfg = dict(fuser="sa", depth=2, num_heads=4, fp_layers=2, fp_heads=4, fp_output_len=1,
                num_classes={"action": K})
    results = {}
    for prec, tol in (("fp32", 1e-3), ("bf16x3", 1e-3), ("bf16", 2e-2)):
        afft_amd.set_precision(prec)
        rt.set_grad_mode("sink")
        torch.manual_seed(0)
        cfg = make_model_cfg(mods, 128, 256, fuser="sa", depth=2, num_heads=4, fp_layers=2, fp_heads=4, T=T)
        model = BaseModel(cfg, {"action": K}, {}).eval()
        state = {k: v.detach().clone() for k, v in model.state_dict().items()}
        data = {m: torch.randn(B, T, C, 1, 1, 1) for m, C in mods.items()}
        tgt = torch.randint(0, K, (B,))
        sub = torch.randint(0, K, (B, T, 1))
        sub[0, :3] = -1
        model = model.to(dev)
        rt.SINK.begin_step()
        out, out_t = model({m: d.to(dev) for m, d in data.items()}, mixup_fn=None, target={"action": tgt.to(dev)},
                           target_subclips={"action": sub.to(dev)}, target_subclips_ignore_index=None)
        losses, _ = BasicLossAccuracy(False)(out, out_t["target"], out_t["target_subclips"])
        total, _ = Runner._reduce_loss(losses, wts, sync=False)
        total.backward()
        rt.SINK.finish_step(list(model.parameters()))
        torch.cuda.synchronize()
        P = {k: v.clone().requires_grad_(True) for k, v in state.items()}
        oout = O.base_model_forward(P, data, ocfg)
        ototal, _ = O.loss(oout, tgt, sub)
        ototal.backward()

        def rel(a, b):
            a, b = a.detach().double().cpu(), b.detach().double()
            return float((a - b).norm() / (b.norm() + 1e-30))

        e_logits = rel(out["logits/action"]["all-fused"], oout["logits/action"]["all-fused"])
        e_past = rel(out["past_logits/action"]["all-fused"], oout["past_logits/action"]["all-fused"])
        e_loss = abs(float(total) - float(ototal)) / max(1.0, abs(float(ototal)))
        gname = "future_predictor.fuser.blocks.0.attn.qkv.weight"
        e_grad = rel(dict(model.named_parameters())[gname].grad, P[gname].grad)
        results[prec] = (e_logits, e_past, e_loss, e_grad)
        print(f"[smoke/{prec}] logits {e_logits:.2e} past_logits {e_past:.2e} loss {e_loss:.2e} grad {e_grad:.2e}")
        assert e_logits < tol and e_past < tol and e_loss < tol and e_grad < max(tol, 1e-3) * 2, (prec, results[prec])
    afft_amd.set_precision("bf16")
    print("smoke OK")
    return results

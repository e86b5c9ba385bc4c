"""Worker of tests/test_model_gpu.py::test_two_ranks_on_one_gpu_*: one data-parallel rank of the REAL HIP path (afft_amd.parallel.Trainer
over the C-ABI library), or -- world size 1 -- the single-process run on the whole batch it is compared with.

    python -m torch.distributed.run --nproc-per-node 2 ... two_rank_gpu.py <case> <precision> <comm_algo> <steps> <out.pt>
    python two_rank_gpu.py <case> <precision> none <steps> <out.pt>

Both ranks sit on cuda:0 and exchange through gloo (RCCL refuses two ranks on one device); everything else is the N > 1 path of
train.py:364-368 as rebuilt here: construction-time broadcast, bucketed reducer on its side stream, per-bucket update (all-reduce form)
or reduce-scatter -> update of the rank's slice -> all-gather of the 16-bit images (sharded form).  Rank 1 starts from perturbed
weights (the broadcast must overwrite them).  After the steps: sync_masters(), an evaluation forward on the FULL batch, and rank 0
saves parameters / momentum / bf16 images by name plus the logits, after checking that rank 1 holds the same bits."""
import os
import sys
from datetime import timedelta

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
for p in (TESTS, os.path.join(TESTS, "golden"), os.path.dirname(TESTS)):
    if p not in sys.path:
        sys.path.insert(0, p)

WTS = {"cls_action": 1.0, "past_cls_action": 1.0, "past_reg": 1.0}


def batch(c, B=4):
    """labels without ignored frames: the mean losses of two half-batches then average to the full-batch mean (the identity DDP relies on)"""
    g = torch.Generator().manual_seed(7)
    data = {m: torch.randn(B, c["T"], C, 1, 1, 1, generator=g) for m, C in c["modal_dims"].items()}
    tgt = torch.randint(0, c["num_classes"], (B,), generator=g)
    sub = torch.randint(0, c["num_classes"], (B, c["T"], 1), generator=g)
    return data, tgt, sub


def main():
    case, precision, algo, steps, out = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if world > 1:
        dist.init_process_group("gloo", timeout=timedelta(seconds=120))
    import afft_amd
    from afft_amd import runtime as rt
    from afft_amd.config import make_model_cfg
    from afft_amd.models.base_model import BaseModel
    from afft_amd.parallel import Trainer
    from helpers import case_tensors
    afft_amd.set_precision(precision)
    rt.set_grad_mode("sink")
    c, state, _, _, _ = case_tensors(case)
    cfg = make_model_cfg(c["modal_dims"], c["d"], c["D"], fuser=c["fuser"], depth=c["depth"], num_heads=c["num_heads"],
                         fp_layers=c["fp_layers"], fp_heads=c["fp_heads"], T=c["T"], drop=0.0)
    model = BaseModel(cfg, num_classes={"action": c["num_classes"]}, class_mappings={})
    model.load_state_dict(state)
    model = model.to(dev).eval()
    if rank == 1:
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.05)
    tr = Trainer(model, WTS, lr=1e-2, momentum=0.9, weight_decay=1e-4, bucket_elems=8192, comm_dtype="fp32",
                 comm_algo=("allreduce" if algo == "none" else algo))
    data, tgt, sub = batch(c)
    B = tgt.shape[0]
    h = B // world
    sl = slice(rank * h, (rank + 1) * h)
    mine = ({m: d[sl].to(dev) for m, d in data.items()}, {"action": tgt[sl].to(dev)}, {"action": sub[sl].to(dev)})
    losses = []
    for _ in range(steps):
        loss, _ = tr.step(*mine)
        losses.append(float(loss))
    torch.cuda.synchronize()
    info = {"buckets": len(tr.reducer.buckets), "split": tr.flat.split, "total": tr.flat.total, "losses": losses}
    if algo == "sharded":
        info["sharded_buckets"] = sum(tr.reducer.sharded_bucket(b) for b in range(len(tr.reducer.buckets)))
        info["stale_before_sync"] = bool(tr.reducer.masters_stale)
        try:
            model.state_dict()
            info["state_dict_on_stale_masters"] = "returned"
        except RuntimeError as ex:
            info["state_dict_on_stale_masters"] = "raised" if "sync_masters" in str(ex) else repr(ex)
    # evaluation forward on the FULL batch: under the sharded update it is also what brings the fp32 masters up to date (every rank)
    with torch.no_grad():
        outs, _ = model({m: d.to(dev) for m, d in data.items()}, mixup_fn=None, target={"action": tgt.to(dev)},
                        target_subclips={"action": sub.to(dev)}, target_subclips_ignore_index=None)
    info["stale_after_eval_forward"] = bool(tr.reducer.masters_stale)
    logits = outs["logits/action"]["all-fused"].float().clone()
    torch.cuda.synchronize()
    ix = tr.flat.index_of()
    named = list(model.named_parameters())
    off = {k: tr.flat.offsets[ix[id(p)]] for k, p in named}
    res = {"params": {k: p.detach().float().cpu().clone() for k, p in named},
           "momentum": {k: tr.opt.buf[off[k]:off[k] + p.numel()].cpu().clone() for k, p in named},
           "images": {k: tr.flat.flat_p16[off[k]:off[k] + p.numel()].cpu().clone() for k, p in named if tr.flat.owns_image(p)},
           "logits": logits.cpu(), "state_dict_keys": sorted(model.state_dict().keys()), "info": info}
    if world > 1:
        both = [None] * world
        dist.all_gather_object(both, res)
        if rank == 0:
            a, b = both[0], both[1]
            same = all(torch.equal(a[g][k], b[g][k]) for g in ("params", "momentum", "images") for k in a[g]) and torch.equal(a["logits"], b["logits"])
            res["info"]["replicas_bitwise_equal"] = bool(same)
    if rank == 0:
        torch.save(res, out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

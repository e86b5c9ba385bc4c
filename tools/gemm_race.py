"""Do the GEMM kernels give the same bits when another stream's GEMMs share the CUs with them?  Every shape below is run alone
(reference bits), then ITER times on the main stream while a second stream keeps launching weight-gradient-shaped GEMMs; any
launch whose output differs from the reference is reported with the tile columns / rows that differ.
usage: python tools/gemm_race.py [ITER]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afft_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
ITER = int(sys.argv[1]) if len(sys.argv) > 1 else 200
g = torch.Generator().manual_seed(3)


def rnd(*shape):
    return (torch.randn(*shape, generator=g) * 0.5).to(torch.bfloat16).to(dev)


R = 5120
# (label, a, b, a_t, b_t, out shape, out dtype): the data-gradient and forward GEMMs of one SA-Fuser block
cases = [
    ("dgrad proj NN 5120x2048x2048", rnd(R, 2048), rnd(2048, 2048), False, False, (R, 2048), torch.bfloat16),
    ("dgrad qkv  NN 5120x2048x6144", rnd(R, 6144), rnd(6144, 2048), False, False, (R, 2048), torch.bfloat16),
    ("dgrad fc2  NN 5120x8192x2048", rnd(R, 2048), rnd(2048, 8192), False, False, (R, 8192), torch.bfloat16),
    ("dgrad fc1  NN 5120x2048x8192", rnd(R, 8192), rnd(8192, 2048), False, False, (R, 2048), torch.bfloat16),
    ("fwd  qkv   NT 5120x6144x2048", rnd(R, 2048), rnd(6144, 2048), False, True, (R, 6144), torch.bfloat16),
    ("fwd  fc2   NT 5120x2048x8192", rnd(R, 8192), rnd(2048, 8192), False, True, (R, 2048), torch.float32),
]
# what the second stream runs: TN weight gradients (fp32 out), K = 1024 (the last block's MLP) and K = 5120
side = [(rnd(1024, 8192), rnd(1024, 2048), (8192, 2048)), (rnd(1024, 2048), rnd(1024, 8192), (2048, 8192)),
        (rnd(R, 2048), rnd(R, 2048), (2048, 2048)), (rnd(R, 6144), rnd(R, 2048), (6144, 2048))]
side_out = [torch.empty(s, dtype=torch.float32, device=dev) for _, _, s in side]
aux = torch.cuda.Stream()
HBM = os.environ.get("HBM", "1") == "1"       # the second stream also streams 2 x 512 MiB through HBM between its GEMMs
COLD = os.environ.get("COLD", "0") == "1"
flush = torch.zeros(256 << 20, dtype=torch.float32, device=dev) if COLD else None
big = torch.zeros(128 << 20, dtype=torch.float32, device=dev) if HBM else None
big2 = torch.zeros(128 << 20, dtype=torch.float32, device=dev) if HBM else None
main = torch.cuda.current_stream()
side_ref = []
for (a, b, s), o in zip(side, side_out):
    ops.gemm(a, b, o, a_t=True)
    side_ref.append(o.clone())
torch.cuda.synchronize()

for label, a, b, a_t, b_t, oshape, odt in cases:
    ref = torch.empty(oshape, dtype=odt, device=dev)
    ops.gemm(a, b, ref, a_t=a_t, b_t=b_t)
    torch.cuda.synchronize()
    outs = [torch.empty(oshape, dtype=odt, device=dev) for _ in range(8)]
    bad_main = bad_side = 0
    where = None
    for it in range(0, ITER, 8):
        with torch.cuda.stream(aux):
            for rep in range(6):
                for (sa, sb, _), so in zip(side, side_out):
                    ops.gemm(sa, sb, so, a_t=True)
                    if HBM:
                        big2.copy_(big)
        for o in outs:
            if COLD:
                flush.add_(1.0)       # operands come from HBM, not from L2 / MALL
            ops.gemm(a, b, o, a_t=a_t, b_t=b_t)
        torch.cuda.synchronize()
        for o in outs:
            if not torch.equal(o, ref):
                bad_main += 1
                if where is None:
                    d = (o != ref)
                    cols = d.any(0).nonzero().flatten()
                    rows = d.any(1).nonzero().flatten()
                    where = (int(d.sum()), int(rows.min()), int(rows.max()), int(cols.min()), int(cols.max()),
                             float((o.float() - ref.float()).abs().max()))
        for so, sr in zip(side_out, side_ref):
            if not torch.equal(so, sr):
                bad_side += 1
    print(f"{label}: {bad_main} of {ITER} main-stream launches differ {where}; side-stream mismatches {bad_side}", flush=True)

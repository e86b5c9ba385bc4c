"""cProfile of the host side of a few training steps (where does the enqueue time go?). usage: host_profile.py [config] [batch]"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
import afft_amd
from afft_amd.parallel import Trainer
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
afft_amd.set_precision("bf16")
dev = torch.device("cuda:0")
model, c = B.build_model(name, dev)
feats, tgt, sub = B.make_inputs(c, batch, c["T"], 0, dev)
tr = Trainer(model, {"cls_action": 1.0, "past_cls_action": 1.0, "past_reg": 1.0})
model.train()
for _ in range(3):
    tr.step(feats, tgt, sub)
torch.cuda.synchronize()
pr = cProfile.Profile()
with torch.autograd.set_multithreading_enabled(False):     # backward in this thread, so that the profiler sees it
    for _ in range(2):
        tr.step(feats, tgt, sub)
    torch.cuda.synchronize()
    pr.enable()
    for _ in range(5):
        tr.step(feats, tgt, sub)
    pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(32)

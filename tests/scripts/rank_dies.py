"""Worker for tests/test_parallel_cpu.py::test_launcher_reports_a_dead_or_hung_rank: two gloo ranks step a GradReducer; in step 2
rank 1 dies (mode 'die': os._exit in the middle of backward) or stops answering (mode 'hang': sleeps forever before its
collective).  Rank 0 must not wait forever: its collective raises (gloo notices the closed connection / the process-group
timeout passes) and the launcher that started both turns that into a non-zero exit."""
import os
import sys
import time
from datetime import timedelta

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from afft_amd import runtime as rt  # noqa: E402
from afft_amd.parallel import FlatParams, GradReducer  # noqa: E402

mode = sys.argv[1]
rank = int(os.environ["RANK"])
dist.init_process_group("gloo", timeout=timedelta(seconds=8))
torch.manual_seed(0)
model = torch.nn.Sequential(torch.nn.Linear(24, 40), torch.nn.Tanh(), torch.nn.Linear(40, 8))
flat = FlatParams(model)
red = GradReducer(flat, bucket_elems=256)
params = list(model.parameters())
for step in range(4):
    red.begin_step()
    flat.flat_g.zero_()
    torch.nn.functional.mse_loss(model(torch.randn(4, 24)), torch.randn(4, 8)).backward()
    if step == 2 and rank == 1:
        if mode == "die":
            os._exit(17)
        time.sleep(3600)
    for p in reversed(params):
        rt.SINK.touched[id(p)] = True
        if rt.SINK.on_grad_ready is not None:
            rt.SINK.on_grad_ready(p)
    red.finish_step()
if rank == 0:
    print('{"ok": true}')
dist.destroy_process_group()

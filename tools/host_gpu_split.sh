#!/bin/bash
# Slow-mode hunt (round 6): the same build gives ~11.7 or ~14.7 ms/step, per PROCESS, on one box, GPU-side (tools/host_gpu_split.py).
# -> gpurun_out/r06_host_gpu_split.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
out=gpurun_out/r06_host_gpu_split.txt; : > $out
for i in 1 2 3 4 5 6; do
  python tools/host_gpu_split.py cfg2 64 default$i 2>&1 | grep "step" | cut -c1-330 | tee -a $out
  GPU_MAX_HW_QUEUES=1 python tools/host_gpu_split.py cfg2 64 hwq1_$i 2>&1 | grep "step" | cut -c1-330 | tee -a $out
  GPU_MAX_HW_QUEUES=8 python tools/host_gpu_split.py cfg2 64 hwq8_$i 2>&1 | grep "step" | cut -c1-330 | tee -a $out
done

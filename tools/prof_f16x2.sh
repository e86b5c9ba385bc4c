#!/bin/bash
# rocprofv3 kernel stats of the cfg2 step in the fp16x2 (parity-grade) precision.  usage (GPU box): bash tools/prof_f16x2.sh <tag>
tag=${1:-prof_f16x2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$tag

timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -o k -- python bench.py --precision fp16x2 --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode --no-reference-loop --no-power --no-ek100 --no-small-batch --no-separate-update > gpurun_out/$tag/stats.log 2>&1
find gpurun_out/$tag -name "*kernel_trace.csv" -delete
tail -c 400 gpurun_out/$tag/stats.log
head -30 gpurun_out/$tag/*/k_kernel_stats.csv 2>/dev/null || head -30 gpurun_out/$tag/k_kernel_stats.csv

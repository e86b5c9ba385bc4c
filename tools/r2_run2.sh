#!/bin/bash
# round 2, GPU call 2: default bench line (new keys), timelines with / without the wgrad CU cap and without overlap
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
( time timeout 900 python bench.py ) > gpurun_out/r2_bench_default.log 2>&1
tail -c 1500 gpurun_out/r2_bench_default.log
bash tools/prof_timeline.sh r2_cap0 --no-parity-mode
bash tools/prof_timeline.sh r2_cap128 --no-parity-mode --wgrad-wgs 128
AFFT_OVERLAP_WGRAD=0 bash tools/prof_timeline.sh r2_serial --no-parity-mode
for t in r2_cap0 r2_cap128 r2_serial; do find gpurun_out/prof_$t -name "*kernel_trace.csv" -delete; find gpurun_out/prof_$t -name "*.db" -delete; done
ls gpurun_out | tail

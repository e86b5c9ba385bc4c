#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/prof_timeline.sh r3f --no-parity-mode
AFFT_OVERLAP_WGRAD=0 bash tools/prof_timeline.sh r3f_serial --no-parity-mode
bash tools/prof_timeline.sh r3f_fullrows --no-parity-mode --full-rows
bash tools/prof_timeline.sh r3f_ek100 --no-parity-mode --config ek100
python tools/gemm_insitu.py cfg2 64 > gpurun_out/r3f_gemm_in_step.txt 2>&1
AFFT_OVERLAP_WGRAD=0 python tools/gemm_insitu.py cfg2 64 > gpurun_out/r3f_gemm_in_step_serial.txt 2>&1
python tools/gemm_insitu.py ek100 64 > gpurun_out/r3f_gemm_in_step_ek100.txt 2>&1
rm -rf gpurun_out/prof_r3f gpurun_out/prof_r3f_serial gpurun_out/prof_r3f_fullrows gpurun_out/prof_r3f_ek100
python tools/timeline_txt.py "default (two streams, token-0 rows in the last fuser block)" gpurun_out/timeline_r3f.json "AFFT_OVERLAP_WGRAD=0: weight gradients on the main stream (every kernel alone)" gpurun_out/timeline_r3f_serial.json "--full-rows (the reference's row set)" gpurun_out/timeline_r3f_fullrows.json "EK100-faithful widths (d = 1024)" gpurun_out/timeline_r3f_ek100.json > gpurun_out/r3f_timelines.txt
head -30 gpurun_out/r3f_timelines.txt

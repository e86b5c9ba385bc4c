#!/bin/bash
# round 3 starting point: stream timelines (overlapped / serial weight gradients), in-step GEMM table, audit test
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -x -q -k "audit" 2>&1 | tail -5 > gpurun_out/r3_03_audit.txt
bash tools/prof_timeline.sh r3a --no-parity-mode
AFFT_OVERLAP_WGRAD=0 bash tools/prof_timeline.sh r3a_serial --no-parity-mode
python tools/gemm_insitu.py cfg2 64 > gpurun_out/r3_gemm_in_step_a.txt 2>&1
AFFT_OVERLAP_WGRAD=0 python tools/gemm_insitu.py cfg2 64 > gpurun_out/r3_gemm_in_step_a_serial.txt 2>&1
rm -rf gpurun_out/prof_r3a gpurun_out/prof_r3a_serial
cat gpurun_out/r3_03_audit.txt

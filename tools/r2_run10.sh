#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r2_t10.log; tail -3 gpurun_out/r2_t10.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2_smoke10.log 2>&1; tail -5 gpurun_out/r2_smoke10.log
bash tools/profile_round.sh r02b

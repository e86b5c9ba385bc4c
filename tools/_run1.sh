#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_model_gpu.py -q -x -k "packed_weight or state_dict_load or projection_on_token or dropin or reference_loop" > gpurun_out/tail_tests.log 2>&1; grep -E "passed|failed|Fatal Python" gpurun_out/tail_tests.log | tail -2
bash tools/profile_round.sh r05e > gpurun_out/profile_round.log 2>&1
tail -c 400 gpurun_out/r05e/bench.json

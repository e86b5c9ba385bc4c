#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r05f > gpurun_out/profile_round.log 2>&1
tail -c 300 gpurun_out/r05f/bench.json

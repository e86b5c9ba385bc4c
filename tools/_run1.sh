#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 2300 python -m pytest tests -q -x -m gpu > gpurun_out/suite.log 2>&1; grep -E "passed|failed|Fatal Python" gpurun_out/suite.log | tail -4

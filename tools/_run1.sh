#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 2300 python -m pytest tests -q -x -m gpu > gpurun_out/suite.log 2>&1; grep -E "passed|failed|Fatal Python" gpurun_out/suite.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2

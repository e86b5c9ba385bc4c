cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05a
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r05a/suite.log
cat gpurun_out/r05a/suite.log

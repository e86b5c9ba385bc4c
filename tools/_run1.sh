cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05d
( timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 ) > gpurun_out/r05d/suite.log
cat gpurun_out/r05d/suite.log
bash tools/profile_round.sh r05d 2>&1 | tail -5

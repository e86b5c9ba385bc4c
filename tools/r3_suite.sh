#!/bin/bash
# the whole GPU suite N times (-x), each run's summary line kept: tools/r3_suite.sh <tag> <N>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; n=${2:-1}
for i in $(seq 1 $n); do
  python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_run$i.log 2>&1
  echo "run $i rc=$? $(grep -E '^[0-9]+ (passed|failed)|passed|failed' gpurun_out/${tag}_run$i.log | tail -1)" | tee -a gpurun_out/${tag}_summary.txt
  grep -E "^(FAILED|ERROR)" gpurun_out/${tag}_run$i.log | head -5 | tee -a gpurun_out/${tag}_summary.txt
  tail -c 3000 gpurun_out/${tag}_run$i.log > gpurun_out/${tag}_run$i.tail; rm gpurun_out/${tag}_run$i.log
done
hostname >> gpurun_out/${tag}_summary.txt

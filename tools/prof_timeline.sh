#!/bin/bash
# rocprofv3 kernel trace of a short bench run + stream timeline summary. usage: tools/prof_timeline.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_$tag
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o tl -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode --no-reference-loop --no-power --no-ek100 "$@" > gpurun_out/prof_$tag.log 2>&1
grep '"metric"' gpurun_out/prof_$tag.log | cut -c1-260
python tools/timeline.py gpurun_out/prof_$tag mse gpurun_out/timeline_$tag.json > /dev/null

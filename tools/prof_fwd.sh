#!/bin/bash
# rocprofv3 kernel trace of eval-mode forwards + stream timeline (marker: the once-per-forward token assembly). usage: tools/prof_fwd.sh <tag> [config] [batch]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_$tag
python tools/fwd_loop.py "$@" > gpurun_out/prof_$tag.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o tl -- python tools/fwd_loop.py "$@" >> gpurun_out/prof_$tag.log 2>&1
python tools/timeline.py gpurun_out/prof_$tag assemble gpurun_out/timeline_$tag.json > gpurun_out/timeline_$tag.txt

#!/bin/bash
# rocprofv3 kernel stats of one bench configuration / precision / batch.  usage (GPU box): bash tools/prof_config.sh <tag> <config> <precision> <batch>
tag=${1:-prof_cfg}; cfg=${2:-ek100}; prec=${3:-bf16}; batch=${4:-64}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$tag
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -o k -- python bench.py --config $cfg --precision $prec --batch $batch --steps 20 --warmup 5 --no-cpu-baseline --no-parity-mode --no-reference-loop --no-power --no-ek100 --no-small-batch --no-separate-update > gpurun_out/$tag/stats.log 2>&1
find gpurun_out/$tag -name "*kernel_trace.csv" -delete
tail -c 300 gpurun_out/$tag/stats.log

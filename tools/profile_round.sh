#!/bin/bash
# Collects what profiles/ holds for a round: bench JSON (with cpu_baseline), rocprofv3 kernel stats of the same command,
# and the two PMC passes for HBM traffic.  usage (on the GPU box): bash tools/profile_round.sh <tag>
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$tag
if [ -z "$SKIP_BENCH" ]; then ( time timeout 900 python bench.py ) > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err; tail -c 600 gpurun_out/$tag/bench.json; fi
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/stats -o k -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode --no-reference-loop --no-power --no-ek100 --no-small-batch --no-separate-update > gpurun_out/$tag/stats.log 2>&1
find gpurun_out/$tag/stats -name "*kernel_trace.csv" -delete
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/$tag/pmc_fetch -o f -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-parity-mode --no-reference-loop --no-power --no-ek100 --no-small-batch --no-separate-update > gpurun_out/$tag/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/$tag/pmc_write -o w -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-parity-mode --no-reference-loop --no-power --no-ek100 --no-small-batch --no-separate-update > gpurun_out/$tag/pmc_write.log 2>&1
python tools/traffic_summary.py gpurun_out/$tag/pmc_fetch gpurun_out/$tag/pmc_write gpurun_out/$tag/traffic.json cfg2 64 bf16 1 > /dev/null
find gpurun_out/$tag -name "*counter_collection.csv" -delete
ls -R gpurun_out/$tag | head -30

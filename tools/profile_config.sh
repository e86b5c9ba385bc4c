#!/bin/bash
# rocprofv3 record of one bench configuration: kernel stats + the two PMC passes + the per-kernel roofline table
# (tools/kernel_report.py).  usage (on the GPU box): bash tools/profile_config.sh <tag> <config> [extra bench args]
tag=$1; cfg=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/$tag
rm -rf $o; mkdir -p $o
timeout 600 python bench.py --config $cfg --no-cpu-baseline --no-parity-mode --no-reference-loop --no-power --no-ek100 "$@" > $o/bench.json 2> $o/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -o k -- python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode --no-reference-loop --no-power --no-ek100 "$@" > $o/stats.log 2>&1
find $o/stats -name "*kernel_trace.csv" -delete
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/pmc_fetch -o f -- python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-parity-mode --no-reference-loop --no-power --no-ek100 "$@" > $o/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/pmc_write -o w -- python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-parity-mode --no-reference-loop --no-power --no-ek100 "$@" > $o/pmc_write.log 2>&1
python tools/kernel_report.py $o/stats $o/pmc_fetch $o/pmc_write $o/bench.json $o/kernel_report.txt "$cfg: per-kernel HBM bandwidth and MFMA rate against the MI355X roofline" | head -30
python tools/traffic_summary.py $o/pmc_fetch $o/pmc_write $o/traffic.json > /dev/null
cp $(find $o/stats -name "*kernel_stats.csv" | head -1) $o/kernel_stats.csv 2>/dev/null
find $o -name "*counter_collection.csv" -delete
ls $o

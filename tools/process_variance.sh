#!/bin/bash
# How much does the headline step time vary from PROCESS to PROCESS on one box, and which kernels carry the difference?
# (round 6: sequential bench.py processes on one box give ~11.7 or ~14.7 ms for the same build.)
# usage (GPU box): bash tools/process_variance.sh [n]  -> gpurun_out/r06_process_variance.txt + gpurun_out/pv_<i>.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
F="--no-cpu-baseline --no-parity-mode --no-reference-loop --no-ek100 --steps 30 --warmup 8"
out=gpurun_out/r06_process_variance.txt; : > $out
for i in $(seq ${1:-8}); do
  timeout 300 python bench.py $F 2>/dev/null > gpurun_out/pv_$i.json
  python - gpurun_out/pv_$i.json $i <<'PY' | tee -a $out
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
bk = {k.replace("gemm_bf16_", "").replace("_kernel", ""): (v["launches"], v["ms"], v["tflops"]) for k, v in r["by_kernel"].items()}
hk = {k: (v["avg_us_in_step"], v["avg_us_alone"]) for k, v in d["hbm_kernels"].items()}
print(f"run {sys.argv[2]}: step {d['ms_per_step']} fwd_p50 {d['fwd_p50_ms']} separate {d['separate_update']['ms_per_step']} power {d['power'].get('package_power_w_avg')} W sclk {d['power'].get('sclk_mhz_median')} "
      f"dom frac {r['frac']} alone {r['alone']['frac']} K1024 {r['by_k_class']['classes'].get('K=1024', {}).get('avg_us')} us K5120 {r['by_k_class']['classes'].get('K=5120', {}).get('avg_us')} us")
print("   gemm (launches, ms, TF/s):", bk)
print("   hbm kernels (us in step, us alone):", hk)
PY
done

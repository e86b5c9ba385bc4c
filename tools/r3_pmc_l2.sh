#!/bin/bash
# L2 hit / miss counters of the GEMM kernels in the bench step: separate --pmc passes, kernel-trace off
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 -L 2>/dev/null | grep -o -E "(TCC_HIT_sum|TCC_MISS_sum|TCC_REQ_sum|TCC_READ_sum|TCC_EA0_RDREQ_sum|TCC_EA0_RDREQ_32B_sum|TCC_EA0_RDREQ_DRAM_sum|TCC_EA0_WRREQ_sum|TCC_EA0_WRREQ_DRAM_sum|TCC_EA0_RD_UNCACHED_32B_sum|MALL[A-Za-z_0-9]*|TCC_[A-Z0-9_]*MALL[A-Za-z_0-9]*|TCC_BUBBLE_sum|TCC_EA0_RDREQ_IO_CREDIT_STALL_sum)" | sort -u > gpurun_out/r3_pmc_available.txt
cat gpurun_out/r3_pmc_available.txt
args="--steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-parity-mode"
for c in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_DRAM_sum" "TCC_REQ_sum TCC_READ_sum"; do
  tag=$(echo $c | tr ' ' '+')
  rm -rf gpurun_out/pmc_l2_$tag
  timeout 600 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_l2_$tag -o p -- python bench.py $args > gpurun_out/pmc_l2_$tag.log 2>&1
  echo "== $c rc=$?" >> gpurun_out/r3_pmc_l2.txt
  python tools/pmc_summary.py gpurun_out/pmc_l2_$tag 2>/dev/null | grep -A4 "gemm_bf16" >> gpurun_out/r3_pmc_l2.txt
  find gpurun_out/pmc_l2_$tag -name "*counter_collection.csv" -delete
done
cat gpurun_out/r3_pmc_l2.txt | head -120

#!/bin/bash
# The race net's self-check: the same stress test against (a) the product library and (b) a control library (built HERE,
# never shipped with the product: afft_amd/lib/libafft_hip_lead7.so is git- and gpurun-ignored until `build` makes it) whose 256x256
# kernel is the general one on every shape (-DAFFT_PP2=0) built with the round-3 schedule, LEAD = 7 (a write-after-read race on a ring slot, profiles/r04_pp_war_race.txt).
#   build here (no GPU needed):  bash tools/race_net.sh build
#   run on the GPU box:          bash tools/race_net.sh run [repeats]      -> gpurun_out/race_net.txt
set -e
cd "$(dirname "$0")/.."
if [ "$1" = "build" ]; then
  cd afft_amd/csrc
  mkdir -p build_var
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DAFFT_PP_LEAD=7 -DAFFT_PP_ALLOW_RACY_LEAD -DAFFT_PP2=0 -c gemm_pp.hip -o build_var/gemm_pp_lead7.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libafft_hip_lead7.so build_var/gemm_pp_lead7.o build/gemm.o build/gemm_bd.o build/norm.o build/attention.o build/attention_mfma.o build/loss.o build/elementwise.o build/sublayer.o
  ls -la ../lib/
  exit 0
fi
n=${2:-5}
mkdir -p gpurun_out
out=gpurun_out/race_net.txt
: > $out
for lib in product lead7; do
  red=0
  for i in $(seq $n); do
    if [ $lib = lead7 ]; then export AFFT_LIB=$PWD/afft_amd/lib/libafft_hip_lead7.so; else unset AFFT_LIB; fi
    if timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "lds_ring" > /tmp/race_${lib}_$i.log 2>&1; then r=green; else r=RED; red=$((red+1)); fi
    echo "$lib run $i: $r $(grep -E 'passed|failed' /tmp/race_${lib}_$i.log | tail -1)" | tee -a $out
    grep -E "elements differed" /tmp/race_${lib}_$i.log | head -4 | cut -c1-200 >> $out || true
  done
  echo "$lib: $red of $n runs red" | tee -a $out
done

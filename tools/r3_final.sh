#!/bin/bash
# end-of-round verification: smoke, the whole GPU suite N times, every configuration's throughput
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; n=${2:-2}
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.txt 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/${tag}_summary.txt; tail -2 gpurun_out/${tag}_smoke.txt
bash tools/r3_suite.sh $tag $n
bash tools/cfg_sweep.sh --no-parity-mode 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${tag}_configs.txt
CFGS="cfg2" bash tools/cfg_sweep.sh --no-parity-mode --batch 16 2>&1 | grep -v amdgpu.ids | sed 's/^/B=16 /' | tee -a gpurun_out/${tag}_configs.txt
CFGS="cfg2" bash tools/cfg_sweep.sh --no-parity-mode --batch 128 2>&1 | grep -v amdgpu.ids | sed 's/^/B=128 /' | tee -a gpurun_out/${tag}_configs.txt
CFGS="cfg1" bash tools/cfg_sweep.sh --no-parity-mode --batch 4 2>&1 | grep -v amdgpu.ids | sed 's/^/B=4 /' | tee -a gpurun_out/${tag}_configs.txt

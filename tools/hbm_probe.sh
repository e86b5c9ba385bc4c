cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for i in 1 2 3 4 5 6; do python tools/hbm_probe.py 6 2>&1 | tail -1 | tee -a gpurun_out/r06_hbm_probe.txt; python tools/host_gpu_split.py cfg2 64 step$i 2>&1 | grep step | cut -c1-200 | tee -a gpurun_out/r06_hbm_probe.txt; done

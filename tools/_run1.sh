cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05a
( time timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "lds_ring" ) 2>&1 | tail -8 > gpurun_out/r05a/race_product.log
cat gpurun_out/r05a/race_product.log
bash tools/race_net.sh run 4 2>&1 | tail -30

cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05a
timeout 1500 python -m pytest tests/test_model_gpu.py -x -q -s -k "fp16x2" 2>&1 | grep -v "^$" | tail -60 > gpurun_out/r05a/m.log
cat gpurun_out/r05a/m.log

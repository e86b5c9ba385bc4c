#!/bin/bash
# the four-wave 256x256 kernels (EXPERIMENTAL=1 build: afft_amd/lib/libafft_hip_exp.so) against the ping-pong kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
AFFT_LIB=$GRAFT_REPO_ROOT/afft_amd/lib/libafft_hip_exp.so VARIANTS=3,5,6 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids

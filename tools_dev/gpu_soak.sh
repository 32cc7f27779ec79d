#!/bin/bash
# soak on the GPU box: usage gpu_soak.sh [frames fused] [frames general]
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python3 tools_dev/soak_fullsize.py ${1:-16384} 1200000 2>&1 | tail -1 | tee gpurun_out/soak_fullsize.log
timeout 1200 python3 tools_dev/soak_general.py ${2:-8192} 1300000 2>&1 | tail -1 | tee gpurun_out/soak_general.log

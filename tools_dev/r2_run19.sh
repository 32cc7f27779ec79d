#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python3 tools_dev/soak_general.py 4096 700000 2>&1 | tail -4 | tee gpurun_out/soak_general.log
timeout 1200 python3 tools_dev/soak_fullsize.py 4096 800000 2>&1 | tail -3 | tee gpurun_out/soak_fullsize.log

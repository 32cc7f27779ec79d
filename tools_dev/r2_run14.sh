#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
bash tools_dev/r2_variants.sh "" "-DRS_GLOBAL_LIST" "" "-DRS_GLOBAL_LIST"
timeout 1200 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "bench_configuration" 2>&1 | tail -2

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
export AMD_LOG_LEVEL=1
timeout 120 python3 -m pytest tests/test_gpu_frontend.py -x -q -m gpu -k "reference_style" 2>&1 | tail -15 | cut -c1-300
echo "---- fused small"
timeout 120 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden_stage or fused" 2>&1 | tail -8 | cut -c1-300
echo "---- bench"
timeout 120 python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-verify --pipeline 1 2>&1 | tail -3 | cut -c1-300

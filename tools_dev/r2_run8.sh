#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python3 tools_dev/loader_trace.py 4 2>&1 | tail -6
timeout 900 python3 tools_dev/loader_bench.py 16 16 > gpurun_out/loader.log 2>&1; cat gpurun_out/loader.log | tail -8
timeout 600 python3 -m pytest tests/test_gpu_frontend.py -x -q -m gpu -k "streaming or cli" 2>&1 | tail -2

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python3 tools_dev/loader_bench.py 16 16 > gpurun_out/loader.log 2>&1; cat gpurun_out/loader.log | tail -9

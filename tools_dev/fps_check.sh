#!/bin/bash
# after a change to the FPS kernels: parity tests that involve FPS, then the serial kernel time
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python3 -m pytest tests -x -q -m gpu -k "fps or fullsize or golden or fused or bench_configuration" 2>&1 | grep -E "passed|failed|error" | tail -3
timeout 300 bash tools_dev/kstats.sh --steps 10 --warmup 2 --no-verify 2>&1 | grep -E "fps_|sum of"
timeout 300 python3 bench.py --cpu-sample 0 --steps 100 2>/dev/null | tail -1 | cut -c90-200

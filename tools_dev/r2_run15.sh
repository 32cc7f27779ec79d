#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests -x -q -m gpu -k "plane or feature or nonuniform or salience or pins or general or fused or frontend or matrix" 2>&1 | tail -4
bash tools_dev/kstats.sh --config 2 --steps 10 --warmup 2 --no-verify 2>&1 | tee gpurun_out/kstats_c2.log
python3 bench.py --config 2 2>/dev/null | tail -1 | cut -c1-400

#!/bin/bash
# full GPU suite + serial kernel stats + both bench lines, every command under its own timeout
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/t_all.log 2>&1; echo "rc=$?" >> gpurun_out/t_all.log
grep -E "passed|failed|error|rc=" gpurun_out/t_all.log | tail -5
timeout 300 bash tools_dev/kstats.sh --steps 10 --warmup 2 --no-verify 2>&1 | grep -v rocclr | head -12
timeout 300 python3 bench.py 2>/dev/null | tail -1 | cut -c90-200
timeout 300 python3 bench.py --config 2 2>/dev/null | tail -1 | cut -c90-200

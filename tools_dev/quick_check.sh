#!/bin/bash
# developer helper (GPU box): the loop of a kernel change -- GPU tests, the default bench line, per-kernel serial durations and
# the wave-level VALU instruction count per kernel.   usage: tools_dev/quick_check.sh [notest] [pytest -k expression]
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
if [ "$1" != "notest" ]; then
  if [ -n "$1" ]; then timeout 1200 python3 -m pytest tests -m gpu -x -q -k "$1" 2>&1 | tail -4
  else timeout 1200 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4; fi
fi
for i in 1 2; do timeout 300 python3 bench.py --no-secondary --cpu-sample 0 --steps 50 2>/dev/null | tail -1 | python3 -c "
import sys, json
r = json.loads(sys.stdin.read()); print('pipelined: %.0f frames/s, %.4f ms/step, verified %s, fps launch %.3f ms' % (r['value'], r['ms_per_step'], r['verified'], r['roofline']['dominant_kernel']['launch_ms']))"; done
bash tools_dev/kstats.sh --steps 10 --warmup 3 --no-verify 2>&1 | grep -v "^{" 
bash tools_dev/pmc.sh "SQ_INSTS_VALU" 2>&1 | grep -v "^{"

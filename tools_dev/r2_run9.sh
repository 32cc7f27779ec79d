#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
for d in 2 3 4 5; do
  python3 bench.py --pipeline $d --cpu-sample 0 --no-verify --steps 60 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('depth $d:', d['value'], d['ms_per_step'], d['roofline']['launch_ms'])"
done
for d in 3 4; do
  python3 bench.py --config 2 --pipeline $d --cpu-sample 0 --no-verify --steps 40 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config2 depth $d:', d['value'], d['ms_per_step'])"
done

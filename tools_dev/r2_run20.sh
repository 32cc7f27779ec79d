#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
for d in 2 3 4 6; do
  echo "c1 depth $d: $(python3 bench.py --cpu-sample 0 --no-verify --steps 100 --pipeline $d 2>/dev/null | tail -1 | cut -c90-135)"
  echo "c2 depth $d: $(python3 bench.py --config 2 --cpu-sample 0 --no-verify --steps 100 --pipeline $d 2>/dev/null | tail -1 | cut -c90-135)"
done

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
for k in 5 10 20 50 200; do
  echo "steps $k: $(timeout 300 python3 bench.py --no-secondary --gpus 1 --steps $k --warmup 5 --cpu-sample 0 --no-verify 2>/dev/null | tail -1 | cut -c98-126)"
done
echo "driver-like: $(timeout 300 python3 bench.py --no-secondary --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c98-126)"

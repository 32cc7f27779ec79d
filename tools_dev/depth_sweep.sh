#!/bin/bash
# developer helper (GPU box): batches in flight x hardware queues (GPU_MAX_HW_QUEUES; the runtime's default is 4)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
for q in "" 2 4 8 16; do for depth in 2 3 4 5 6 8; do
  GPU_MAX_HW_QUEUES=$q timeout 300 python3 bench.py --no-secondary --cpu-sample 0 --no-verify --steps 100 --pipeline $depth "$@" 2>/dev/null | tail -1 | python3 -c "
import sys, json
r = json.loads(sys.stdin.read()); print('GPU_MAX_HW_QUEUES=%-3s depth $depth %8.0f frames/s  %.4f ms/step  fps launch %.3f ms' % ('$q', r['value'], r['ms_per_step'], r['roofline']['dominant_kernel']['launch_ms']))"
done; done

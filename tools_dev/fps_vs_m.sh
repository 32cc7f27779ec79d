#!/bin/bash
# FPS kernel duration against the number of centres: where the 99 dependent iterations spend their time
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
for m in 2 4 8 16 24 50 100; do
  echo "M=$m $(timeout 300 bash tools_dev/kstats.sh --steps 6 --warmup 2 --no-verify --clusters $m 2>&1 | grep fps_regtab)"
done | tee gpurun_out/fps_vs_m.log

#!/bin/bash
# A/B of compile-time variants on ONE box: usage ab_bench.sh "<flags A>" "<flags B>" ...  (each variant: 3 pipelined bench runs)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
for rep in 1 2; do
for f in "$@"; do
  RPCC_EXTRA_FLAGS="$f" python3 -c "
import sys; sys.path.insert(0,'.')
import rpcc_amd
from rpcc_amd import build as b
b.build(force=True)" 2>&1 | grep -E " error" -A3
  for i in 1 2; do
    echo "[$f] $(timeout 300 python3 bench.py --cpu-sample 0 --no-verify --steps 100 $BARGS 2>/dev/null | tail -1 | cut -c98-130)"
  done
done
done

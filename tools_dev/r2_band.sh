#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
for f in "-DBAND_EXP_NOATOMIC" "-DBAND_EXP_NOATOMIC -DBAND_INFLIGHT=8" "-DBAND_THREADS=512 -DBAND_INFLIGHT=8"; do
  echo "== $f"; RPCC_EXTRA_FLAGS="$f" bash tools_dev/kstats.sh --steps 6 --warmup 2 --no-verify 2>&1 | grep -E "band|sum of"
done

#!/bin/bash
# usage: tools_dev/variants.sh "<flags>" ... : rebuild with the flags, serial kernel stats (grep pattern in $PAT)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
for f in "$@"; do
  echo "== $f"
  RPCC_EXTRA_FLAGS="$f" python3 -c "
import sys; sys.path.insert(0,'.')
import rpcc_amd
from rpcc_amd import build as b
b.build(force=True)"
  bash tools_dev/kstats.sh --steps 6 --warmup 2 --no-verify $KARGS 2>&1 | grep -E "${PAT:-assign|sum of}"
done

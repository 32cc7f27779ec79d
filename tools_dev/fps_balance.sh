#!/bin/bash
# developer helper (GPU box): DEVTRACE build, then the FPS visit statistics and the per-phase cycles (B = 256 and B = 8 frames)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
RPCC_EXTRA_FLAGS="-DRPCC_DEVTRACE" python3 -c "
import sys; sys.path.insert(0,'.')
import rpcc_amd
from rpcc_amd import build as b
b.build(force=True)" 2>&1 | grep -E "error" -A3
timeout 300 python3 tools_dev/fps_balance.py 2>&1 | tail -12
timeout 300 python3 tools_dev/fps_phases.py 256 2>&1 | tail -12
timeout 300 python3 tools_dev/fps_phases.py 8 2>&1 | tail -12

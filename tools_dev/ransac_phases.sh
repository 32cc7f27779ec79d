cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
RPCC_EXTRA_FLAGS="-DRPCC_DEVTRACE" python3 -c "
import sys; sys.path.insert(0,'.')
import rpcc_amd
from rpcc_amd import build as b
b.build(force=True)" 2>&1 | grep -E "error" -A3
timeout 300 python3 tools_dev/phase_times.py 256 2>&1 | grep -A12 "^ransac"

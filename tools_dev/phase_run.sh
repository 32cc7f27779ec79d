#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
timeout 300 python3 tools_dev/phase_times.py 256 2>&1 | grep -v "^$" | tail -30

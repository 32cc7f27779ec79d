#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
timeout 600 python3 tools_dev/loader_bench.py 16 32 2>&1 | grep -v "^$" | tail -12

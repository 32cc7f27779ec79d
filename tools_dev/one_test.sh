#!/bin/bash
# usage: one_test.sh "<pytest -k expression>" [file]
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
timeout 900 python3 -m pytest ${2:-tests} -x -q -m gpu -k "$1" 2>&1 | grep -vE "^(HIP|ROCm|Hostname|Librccl|RCCL|/opt)" | tail -25

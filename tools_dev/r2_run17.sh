#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 2400 python3 -m pytest tests -x -q -m gpu > gpurun_out/t_all.log 2>&1; echo "rc=$?" >> gpurun_out/t_all.log
grep -E "passed|failed|error|rc=" gpurun_out/t_all.log | tail -5

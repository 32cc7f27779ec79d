#!/bin/bash
# serial per-kernel durations for several library variants (kstats), kernels of interest only
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
for l in "$@"; do echo "== $l"; RPCC_HIP_LIB=$PWD/r-pcc_amd/lib/variants/$l.so bash tools_dev/kstats.sh --steps 10 --warmup 3 --no-verify 2>&1 | grep -E "assign|mask|quant|hist|sum of"; done

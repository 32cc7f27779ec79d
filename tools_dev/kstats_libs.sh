#!/bin/bash
# serial per-kernel durations (kstats.sh) for several libraries given by path, all kernels
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
for l in "$@"; do echo "== $l"; RPCC_HIP_LIB=$PWD/$l bash tools_dev/kstats.sh --steps 10 --warmup 3 --no-verify 2>&1 | grep -v "^{" ; done

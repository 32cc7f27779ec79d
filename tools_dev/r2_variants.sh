#!/bin/bash
# usage: tools_dev/r2_variants.sh "flags1" "flags2" ...   -> gpurun_out/variants.log
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
: > gpurun_out/variants.log
for f in "$@"; do
  RPCC_EXTRA_FLAGS="$f" timeout 600 python3 tools_dev/fps_time.py 2>&1 | grep -E "VARIANT|rror" >> gpurun_out/variants.log
done
cat gpurun_out/variants.log

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
export PAT="plane_model" KARGS="--config 2"
bash tools_dev/r2_band.sh "" "-DPL_SKIP_VALID" "-DRSX_NOREFIT" "-DPL_WAVES=5"  2>&1 | grep -E "^==|plane_model" | tee gpurun_out/plane_exp.log

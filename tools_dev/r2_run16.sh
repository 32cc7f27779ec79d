#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
export PAT="features" KARGS="--config 2"
bash tools_dev/r2_band.sh "-DFEAT_EXP_NOSEL" "-DFEAT_EXP_NOSEL -DFEAT_EXP_NOCURV" "-DFEAT_NO_ROWMODE" 2>&1 | grep -E "^==|features" | tee gpurun_out/feat_exp.log

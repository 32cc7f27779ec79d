#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
export PAT="plane_model|label_order" KARGS="--config 2"
bash tools_dev/r2_band.sh "" "-DRSX_NOSCORE" "-DPL_SKIP_MEAN" "-DPL_SKIP_RANSAC" 2>&1 | grep -E "^==|plane_model|label_order" | tee gpurun_out/plane_exp.log
export KARGS="--config 2 --batch 64"
bash tools_dev/r2_band.sh "" 2>&1 | grep -E "^==|plane_model" | tee -a gpurun_out/plane_exp.log

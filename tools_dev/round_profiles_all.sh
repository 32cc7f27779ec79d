#!/bin/bash
# GPU box: the three committed profile sets of a round (configs[1], configs[2], the real sweep); then, locally:
#   python tools_profiles.py rNN_vM ; python tools_profiles.py rNN_vM_c2 --config 2 --src gpurun_out/c2 ; ... (see profiles/README.md)
cd "${GRAFT_REPO_ROOT:-.}"
bash tools_dev/round_profiles.sh; mkdir -p gpurun_out/c1 && mv gpurun_out/prof_* gpurun_out/c1/
bash tools_dev/round_profiles.sh --config 2; mkdir -p gpurun_out/c2 && mv gpurun_out/prof_* gpurun_out/c2/
bash tools_dev/round_profiles.sh --input tests/golden/example_64E.npz; mkdir -p gpurun_out/real && mv gpurun_out/prof_* gpurun_out/real/
ls gpurun_out/c1 gpurun_out/c2 gpurun_out/real | head -40

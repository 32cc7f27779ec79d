#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
RPCC_EXTRA_FLAGS="-DFPS_PROF2" python3 -c "
import sys; sys.path.insert(0,'.')
import rpcc_amd
from rpcc_amd import build as b
b.build(force=True)"
python3 tools_dev/phase_times.py > gpurun_out/phase2.log 2>&1
cat gpurun_out/phase2.log | tail -12
timeout 900 python3 -m pytest tests/test_gpu_pins.py -x -q -m gpu 2>&1 | tail -5

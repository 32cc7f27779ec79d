#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
python3 -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
timeout 2400 python3 -m pytest tests -x -q -m gpu > gpurun_out/t10.log 2>&1; echo "rc=$?" >> gpurun_out/t10.log
grep -E "passed|failed|rc=|Error|error" gpurun_out/t10.log | tail -5
bash tools_dev/r2_variants.sh ""
timeout 600 bash tools_dev/kstats.sh --steps 10 --warmup 2 --no-verify > gpurun_out/kstats_serial.log 2>&1; head -14 gpurun_out/kstats_serial.log

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
python3 -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "projection or golden or bench_configuration or fused" > gpurun_out/t11.log 2>&1; echo "rc=$?" >> gpurun_out/t11.log
grep -E "passed|failed|rc=|Error|error" gpurun_out/t11.log | tail -3
bash tools_dev/r2_variants.sh ""
timeout 600 bash tools_dev/kstats.sh --steps 10 --warmup 2 --no-verify > gpurun_out/kstats_serial.log 2>&1; head -14 gpurun_out/kstats_serial.log

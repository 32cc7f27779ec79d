#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
python3 -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "fps or fused or golden or bench_configuration or full_size" > gpurun_out/t4.log 2>&1; echo "rc=$?" >> gpurun_out/t4.log
tail -3 gpurun_out/t4.log
bash tools_dev/r2_variants.sh ""
timeout 600 bash tools_dev/kstats.sh --steps 10 --warmup 2 --no-verify > gpurun_out/kstats_serial.log 2>&1; cat gpurun_out/kstats_serial.log

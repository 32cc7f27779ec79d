#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
python3 -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
timeout 600 python3 -m pytest tests/test_gpu_frontend.py -x -q -m gpu -k "streaming" 2>&1 | tail -2
timeout 900 python3 tools_dev/loader_bench.py 12 16 > gpurun_out/loader.log 2>&1; cat gpurun_out/loader.log | tail -8
timeout 600 bash tools_dev/kstats.sh --steps 10 --warmup 2 --no-verify --config 2 > gpurun_out/kstats_c2.log 2>&1; cat gpurun_out/kstats_c2.log

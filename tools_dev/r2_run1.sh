#!/bin/bash
# round-2 GPU session: new full-size tests, whole -m gpu suite, FPS phase stamps, serial kernel stats, bench
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out
python3 -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
timeout 900 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/t_fullsize.log 2>&1; echo "fullsize rc=$?" >> gpurun_out/t_fullsize.log
timeout 1500 python3 -m pytest tests -x -q -m gpu --deselect tests/test_gpu_fullsize.py > gpurun_out/t_gpu.log 2>&1; echo "gpu rc=$?" >> gpurun_out/t_gpu.log
timeout 300 python3 tools_dev/phase_times.py > gpurun_out/phase.log 2>&1
timeout 600 bash tools_dev/kstats.sh --steps 10 --warmup 2 --no-verify > gpurun_out/kstats_serial.log 2>&1
timeout 600 python3 bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
tail -3 gpurun_out/t_fullsize.log gpurun_out/t_gpu.log; cat gpurun_out/phase.log | tail -30; cat gpurun_out/kstats_serial.log; cat gpurun_out/bench.json | cut -c1-600

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
python3 -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
timeout 2400 python3 -m pytest tests -x -q -m gpu > gpurun_out/t5.log 2>&1; echo "rc=$?" >> gpurun_out/t5.log
tail -6 gpurun_out/t5.log
timeout 900 python3 tools_dev/loader_bench.py 12 > gpurun_out/loader.log 2>&1; cat gpurun_out/loader.log | tail -12
timeout 600 python3 bench.py --config 2 > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err; cut -c1-400 gpurun_out/bench_c2.json; tail -3 gpurun_out/bench_c2.err
timeout 600 python3 bench.py --input tests/golden/example_64E.npz > gpurun_out/bench_real.json 2> gpurun_out/bench_real.err; cut -c1-400 gpurun_out/bench_real.json; tail -3 gpurun_out/bench_real.err

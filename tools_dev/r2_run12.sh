#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
python3 -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
timeout 2400 python3 -m pytest tests -x -q -m gpu > gpurun_out/t12.log 2>&1; echo "rc=$?" >> gpurun_out/t12.log
grep -E "passed|failed|rc=|Error|error" gpurun_out/t12.log | tail -4
python3 bench.py --cpu-sample 0 --steps 60 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config1:', d['value'], d['ms_per_step'], d['verified'])"
python3 bench.py --config 2 --cpu-sample 0 --steps 40 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config2:', d['value'], d['ms_per_step'], d['verified'])"

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
python3 -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
for f in "--force-gather" "--force-gather --gather-payloads"; do
  python3 bench.py $f --cpu-sample 0 --steps 30 2> gpurun_out/fg.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$f:', d['value'], d['ms_per_step'], d['verified'], d['n_gpus'], d['config']['exchange_bytes_per_step'], d['config']['sharding'][:120])"
  tail -2 gpurun_out/fg.err
done
WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 python3 bench.py --gpus 1 --cpu-sample 0 --steps 10 2>/dev/null | cut -c1-200
python3 bench.py --gpus 2; echo "rc=$?"

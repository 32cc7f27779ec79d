#!/bin/bash
# developer helper (GPU box): the ordered projection -- tests, phases, serial kernel times, pipelined A/B against the record kernels (RPCC_PROJECT_FLAGS=16: probe + window kernel, 0: record kernels)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ordered.py -x -q 2>&1 | tail -2
RPCC_HIP_LIB=$PWD/r-pcc_amd/lib/variants/trace.so timeout 120 python tools_dev/ordered_phases.py 256
echo "== kstats real (ordered kernel)"; RPCC_PROJECT_FLAGS=16 bash tools_dev/kstats.sh --input tests/golden/example_64E.npz 2>&1 | grep -i "proj\|ransac\|sum"
echo "== bench real: default (ordered) vs records"
for rep in 1 2; do
  for f in 16 0; do
    RPCC_PROJECT_FLAGS=$f timeout 300 python3 bench.py --no-secondary --cpu-sample 0 --steps 100 --input tests/golden/example_64E.npz 2>/dev/null | tail -1 | python3 -c "
import sys, json
r = json.loads(sys.stdin.read()); print('flags $f  %8.0f frames/s  %.4f ms/step  verified %s' % (r['value'], r['ms_per_step'], r['verified']))"
  done
done
if [ "$1" == "headline" ]; then
echo "== bench headline: default vs no probe launch"
for rep in 1 2; do
  for f in 16 0; do
    RPCC_PROJECT_FLAGS=$f timeout 300 python3 bench.py --no-secondary --cpu-sample 0 --steps 100 2>/dev/null | tail -1 | python3 -c "
import sys, json
r = json.loads(sys.stdin.read()); print('flags $f  %8.0f frames/s  %.4f ms/step  verified %s' % (r['value'], r['ms_per_step'], r['verified']))"
  done
done
fi

#!/bin/bash
# developer helper (GPU box): ground-less sweeps -- the chip-wide whole-cloud fit against the library before it (variants/pre_wc.so), alternating runs
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "ground" 2>&1 | tail -3
run() {  # lib, extra args
  RPCC_HIP_LIB=$1 timeout 300 python3 bench.py --no-secondary --cpu-sample 0 --steps 100 ${@:2} 2>/dev/null | tail -1 | python3 -c "
import sys, json
r = json.loads(sys.stdin.read()); print('%-28s %-18s %8.0f frames/s  %.4f ms/step  verified %s' % ('$(basename $1)', '${*:2}', r['value'], r['ms_per_step'], r['verified']))"
}
NEW=$PWD/r-pcc_amd/lib/librpcc_hip.so; OLD=$PWD/r-pcc_amd/lib/variants/pre_wc.so
for rep in 1 2 3; do run $NEW; run $OLD; done
for k in 1 8 64; do for rep in 1 2; do run $NEW --groundless $k; run $OLD --groundless $k; done; done
echo "== serial kernel times, 8 ground-less sweeps"; bash tools_dev/kstats.sh --groundless 8 2>&1 | grep -i "ransac\|ground_wc\|sum"
echo "== serial kernel times, headline"; bash tools_dev/kstats.sh 2>&1 | grep -i "ransac\|ground_wc\|sum"
echo "== mixed lidars (the VLP16 group holds a ground-less sweep)"
timeout 600 python3 tools_dev/mixed_libs.py r-pcc_amd/lib/librpcc_hip.so r-pcc_amd/lib/variants/pre_wc.so 2>/dev/null | grep "slots"

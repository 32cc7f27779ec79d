#!/bin/bash
# developer helper (GPU box): A/B of library builds inside ONE session (box-to-box variation is 3-5 %): alternating runs of the
# default bench command per library.   usage: tools_dev/ab.sh libA.so libB.so ... [-- bench.py args]
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
libs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done; [ "$1" == "--" ] && shift
for rep in 1 2 3; do for l in "${libs[@]}"; do
  RPCC_HIP_LIB=$PWD/$l timeout 300 python3 bench.py --no-secondary --cpu-sample 0 --steps 100 "$@" 2>/dev/null | tail -1 | python3 -c "
import sys, json
r = json.loads(sys.stdin.read()); print('%-44s %8.0f frames/s  %.4f ms/step  verified %s  fps launch %.3f ms' % ('$l', r['value'], r['ms_per_step'], r['verified'], r['roofline']['dominant_kernel']['launch_ms']))"
done; done

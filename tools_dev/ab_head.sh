# developer helper (GPU box): the working tree's library against the last commit's (tools_dev/build_head_variant.sh), alternating bench runs.
# usage: bash tools_dev/ab_head.sh [reps] [steps] [extra bench.py flags]
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
reps=${1:-6}; steps=${2:-200}; shift; shift
for rep in $(seq 1 $reps); do for l in r-pcc_amd/lib/librpcc_hip.so r-pcc_amd/lib/variants/head.so; do
  RPCC_HIP_LIB=$PWD/$l timeout 300 python3 bench.py --no-secondary --cpu-sample 0 --steps $steps "$@" 2>/dev/null | tail -1 | python3 -c "
import sys, json
r = json.loads(sys.stdin.read()); print('%-40s %8.0f frames/s  %.4f ms/step  verified %s' % ('$l', r['value'], r['ms_per_step'], r['verified']))"
done; done

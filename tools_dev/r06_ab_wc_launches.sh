cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
for rep in 1 2 3 4 5 6; do for l in r-pcc_amd/lib/librpcc_hip.so r-pcc_amd/lib/variants/wcs256.so r-pcc_amd/lib/variants/nowc.so; do
  RPCC_HIP_LIB=$PWD/$l timeout 300 python3 bench.py --no-secondary --cpu-sample 0 --steps 200 2>/dev/null | tail -1 | python3 -c "
import sys, json
r = json.loads(sys.stdin.read()); print('%-40s %8.0f frames/s  %.4f ms/step  verified %s' % ('$l', r['value'], r['ms_per_step'], r['verified']))"
done; done

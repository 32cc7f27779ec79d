#!/bin/bash
# developer helper (GPU box): per-kernel durations of one serial bench run -> stdout table
# usage: tools_dev/kstats.sh [bench.py args]
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
rm -rf /tmp/kst && mkdir -p /tmp/kst
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o k -- python3 bench.py --no-secondary --cpu-sample 0 --pipeline 1 "$@" > /tmp/kst/bench.log 2>&1
f=$(find /tmp/kst -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if not any(t in r['Name'] for t in ('at::native', 'rocprim', 'hipcub', 'rocblas'))]
tot = 0
for r in rows:
    n = r['Name'].split('(')[0].replace('void ', '')[:44]; a = float(r['AverageNs']) / 1e3; tot += a
    print("%-46s %5s %9.1f us" % (n, r['Calls'], a))
print("sum of averages %.1f us" % tot)
PY
tail -1 /tmp/kst/bench.log | cut -c1-330

#!/bin/bash
# developer helper (GPU box): per-kernel durations of tools_dev/microbench2.py
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
rm -rf /tmp/kst2 && mkdir -p /tmp/kst2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst2 -o k -- python3 tools_dev/microbench2.py "$@" > /tmp/kst2/log 2>&1
f=$(find /tmp/kst2 -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if not any(t in r['Name'] for t in ('at::native', 'rocprim', 'hipcub', 'rocblas', 'rocclr'))]
for r in rows:
    print("%-46s %5s %9.1f us" % (r['Name'].split('(')[0].replace('void ', '')[:44], r['Calls'], float(r['AverageNs']) / 1e3))
PY

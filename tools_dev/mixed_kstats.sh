#!/bin/bash
# GPU box: per-kernel durations of the mixed-lidar call (bench.run_mixed, SLOTS mixed batches in flight, default 1 = serial)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
rm -rf /tmp/mxk && mkdir -p /tmp/mxk
cat > /tmp/mxk/run.py <<'PY'
import os, sys, types, torch
sys.path.insert(0, os.getcwd())
import rpcc_amd, bench
r = bench.run_mixed(types.SimpleNamespace(accuracy=0.02), dict(dev=torch.device("cuda:0")), per=85, reps=24, slots=int(os.environ.get("SLOTS", "1")))
print(r["value"], r["ms_per_mixed_batch"])
PY
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mxk/out -o k -- python3 /tmp/mxk/run.py > /tmp/mxk/log 2>&1
tail -1 /tmp/mxk/log
f=$(find /tmp/mxk/out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if not any(t in r['Name'] for t in ('at::native', 'rocprim', 'hipcub', 'rocblas', 'rocclr'))]
tot = 0
for r in sorted(rows, key=lambda r: -float(r['AverageNs']) * int(r['Calls'])):
    n = r['Name'].split('(')[0].replace('void ', '')[:52]; a = float(r['AverageNs']) / 1e3; c = int(r['Calls'])
    print("%-54s %5d %9.1f us" % (n, c, a))
PY

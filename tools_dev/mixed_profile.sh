#!/bin/bash
# GPU box: per-kernel durations of the mixed-lidar secondary (85 + 85 + 85 sweeps, three geometry groups on three streams)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
rm -rf /tmp/mx && mkdir -p /tmp/mx
cat > /tmp/mx/run.py <<'PY'
import os, sys, types, torch
sys.path.insert(0, os.getcwd())
import rpcc_amd, bench
r = bench.run_mixed(types.SimpleNamespace(accuracy=0.02), dict(dev=torch.device("cuda:0")), per=int(sys.argv[1]), reps=12, slots=int(os.environ.get("SLOTS", "1")))
print(r["value"], r["ms_per_mixed_batch"])
PY
for per in 85 256; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mx/$per -o k -- python3 /tmp/mx/run.py $per > /tmp/mx/log$per 2>&1
f=$(find /tmp/mx/$per -name "*kernel_stats.csv" | head -1)
echo "== $per frames per geometry: $(tail -1 /tmp/mx/log$per)"
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if not any(t in r['Name'] for t in ('at::native', 'rocprim', 'hipcub', 'rocblas', 'Cijk', 'elementwise'))]
tot = 0
for r in rows[:24]:
    n = r['Name'].split('(')[0].replace('void ', '')[:52]; a = float(r['AverageNs']) / 1e3
    print("%-54s %5s avg %8.1f us  min %8.1f max %8.1f  total %8.1f ms" % (n, r['Calls'], a, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
done

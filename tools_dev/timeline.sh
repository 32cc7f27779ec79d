#!/bin/bash
# developer helper (GPU box): kernel timeline of the pipelined bench (steady state), one line per dispatch
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
rm -rf /tmp/tl && mkdir -p /tmp/tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python3 bench.py --cpu-sample 0 --steps 8 --warmup 3 "$@" > /tmp/tl/bench.log 2>&1
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if not any(t in r['Kernel_Name'] for t in ('at::native', 'rocprim', 'hipcub', 'rocblas', 'rocclr'))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# steady state: last 40% of dispatches
n = len(rows); rows = rows[int(n * 0.55):int(n * 0.85)]
t0 = int(rows[0]['Start_Timestamp'])
qs = sorted({r['Queue_Id'] for r in rows})
for r in rows:
    nm = r['Kernel_Name'].split('(')[0].replace('void ', '')[:24]
    s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
    print("q%d %-24s %9.1f -> %9.1f  (%7.1f us)" % (qs.index(r['Queue_Id']), nm, s, e, e - s))
PY
tail -1 /tmp/tl/bench.log | cut -c100-200

#!/bin/bash
# developer helper (GPU box): kernel trace of the default (pipelined) bench command -> gpurun_out/timeline.csv
# (start / end / queue of every dispatch; analysed locally by tools_dev/timeline.py)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
rm -rf /tmp/tl && mkdir -p /tmp/tl
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python3 bench.py --no-secondary --cpu-sample 0 --no-verify --steps 12 --warmup 3 "$@" > /tmp/tl/bench.log 2>&1
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
python3 - "$f" gpurun_out/timeline.csv <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if not any(t in r['Kernel_Name'] for t in ('at::native', 'rocprim', 'hipcub', 'rocblas', 'rocclr'))]
w = csv.writer(open(sys.argv[2], 'w'))
w.writerow(['kernel', 'queue', 'stream', 'start_ns', 'end_ns', 'wg', 'grid'])
for r in rows:
    w.writerow([r['Kernel_Name'].split('(')[0].replace('void ', '')[:40], r.get('Queue_Id', ''), r.get('Stream_Id', ''), r['Start_Timestamp'], r['End_Timestamp'], r.get('Workgroup_Size', ''), r.get('Grid_Size', '')])
print(len(rows), "dispatches")
PY
tail -1 /tmp/tl/bench.log | cut -c1-200

#!/bin/bash
# GPU box: how many launches of the mixed-lidar secondary run side by side?  Kernel trace of bench.run_mixed with SLOTS mixed batches in
# flight; prints, per hardware queue and per stream, the busy share of the traced region and the mean number of kernels in flight.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
rm -rf /tmp/mxt && mkdir -p /tmp/mxt
cat > /tmp/mxt/run.py <<'PY'
import os, sys, types, torch
sys.path.insert(0, os.getcwd())
import rpcc_amd, bench
r = bench.run_mixed(types.SimpleNamespace(accuracy=0.02), dict(dev=torch.device("cuda:0")), per=85, reps=24, slots=int(os.environ.get("SLOTS", "1")), by_streams=bool(int(os.environ.get("BY_STREAMS", "0"))))
print(r["value"], r["ms_per_mixed_batch"])
PY
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/mxt/out -o k -- python3 /tmp/mxt/run.py > /tmp/mxt/log 2>&1
tail -1 /tmp/mxt/log
f=$(find /tmp/mxt/out -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print("columns:", list(rows[0].keys()))
ours = [r for r in rows if any(t in r['Kernel_Name'] for t in ('fps_regtab', 'ground_ransac', 'plane_model', 'assign_kernel', 'project_', 'features_', 'predict_quantize', 'label_order', 'contour_', 'ground_mask', 'model_', 'salience', 'tile_scan'))]
# the timed region: the last 24 mixed batches = last 24 * 3 FPS launches
fps = sorted((int(r['Start_Timestamp']) for r in ours if 'fps_regtab' in r['Kernel_Name']))
t0 = fps[-(len(fps) * 24 // 30)]   # 24 timed of 30 mixed batches (2 x slots warm-up ones precede them when slots = 3)
sel = [r for r in ours if int(r['Start_Timestamp']) >= t0]
t1 = max(int(r['End_Timestamp']) for r in sel)
wall = t1 - t0
tot = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in sel)
print("region %.2f ms, %d launches, sum of durations %.2f ms -> %.2f kernels in flight on average" % (wall / 1e6, len(sel), tot / 1e6, tot / wall))
for key in ('Queue_Id', 'Stream_Id'):
    if key not in sel[0]:
        continue
    busy = collections.Counter()
    for r in sel:
        busy[r[key]] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    print(key, {k: round(v / wall, 2) for k, v in sorted(busy.items())})
# concurrency histogram
ev = []
for r in sel:
    ev.append((int(r['Start_Timestamp']), 1)); ev.append((int(r['End_Timestamp']), -1))
ev.sort()
hist = collections.Counter(); cur = 0; last = t0
for t, d in ev:
    hist[cur] += t - last; last = t; cur += d
print("time share by number of kernels in flight:", {k: round(v / wall, 3) for k, v in sorted(hist.items())})
per = collections.defaultdict(lambda: [0, 0])
for r in sel:
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:60]
    per[k][0] += 1; per[k][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
print("per mixed batch (24 in the region): launches, total us")
for k, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print("  %-62s %5.1f %8.1f" % (k, n / 24, t / 24e3))
PY

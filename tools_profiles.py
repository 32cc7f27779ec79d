#!/usr/bin/env python3
"""Build the committed profiles/ summaries from raw rocprofv3 CSVs in gpurun_out/ (scratch).
usage: tools_profiles.py <tag>   e.g. r01_v4   (expects p_default_*, p_serial_*, pmc2_* in gpurun_out/)"""
import collections, csv, json, os, sys
tag = sys.argv[1]
G, P = "gpurun_out", "profiles"
SKIP = ('at::native', 'rocblas', 'rocclr', 'rocprim', 'hipcub', 'anonymous')

def stats(src, dst_md, dst_csv):
    rows = list(csv.reader(open(src)))
    keep = [rows[0]] + [r for r in rows[1:] if not any(t in r[0] for t in SKIP)]
    csv.writer(open(dst_csv, 'w')).writerows(keep)
    L = ["| kernel | calls | avg us | min us | max us |", "|---|---|---|---|---|"]
    for r in csv.DictReader(open(dst_csv)):
        L.append("| `%s` | %s | %.1f | %.1f | %.1f |" % (r['Name'].split('(')[0].replace('void ', '')[:60], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
    open(dst_md, 'w').write("\n".join(L) + "\n")
    return {r['Name'].split('(')[0].replace('void ', ''): float(r['AverageNs']) / 1e3 for r in csv.DictReader(open(dst_csv))}

d = stats(f"{G}/p_default_kernel_stats.csv", f"{P}/{tag}_kernel_stats_pipelined.md", f"{P}/{tag}_kernel_stats_pipelined_raw.csv")
s = stats(f"{G}/p_serial_kernel_stats.csv", f"{P}/{tag}_kernel_stats_serial.md", f"{P}/{tag}_kernel_stats_serial_raw.csv")
for n in ("p_default_bench", "p_serial_bench", "bench_final"):
    open(f"{P}/{tag}_{n}.json", 'w').write(open(f"{G}/{n}.json").read().strip().splitlines()[-1] + "\n")

out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f"{G}/pmc2_{c}.csv")):
        n = r['Kernel_Name']
        if any(t in n for t in SKIP) or r['Counter_Name'] != c:
            continue
        agg[n.split('(')[0].replace('void ', '')].append((float(r['Counter_Value']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
    out[c] = agg
names = sorted(out['FETCH_SIZE'], key=lambda n: -sum(v for v, _ in out['FETCH_SIZE'][n]))
L = [f"# {tag} PMC pass: HBM-side traffic per kernel launch (B = 256 frames of 64x2048, serial steps)", "",
     "Two separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md):", "",
     "    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --pipeline 1",
     "    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --pipeline 1", "",
     "Units: the counters are KB.  gfx950 correction (guide, section HBM): FETCH_SIZE reports half of the bytes of a coalesced",
     "stream.  Calibration on kernels with known byte counts: `project_pix_kernel` reads 12 B x 29.0 M points = 348 MB,",
     "`assign_kernel` reads the 134 MB range image; both report about half, so read = 2 x FETCH_SIZE also for 4 B/lane loads.",
     "WRITE_SIZE needs no correction (`project_pix_kernel` writes 232 MB of records, `assign_kernel` 33.5 MB of labels).", "",
     "| kernel | launches | FETCH_SIZE raw MB | read MB (x2) | WRITE_SIZE MB | HBM-side traffic MB/launch | avg us (profiled) |", "|---|---|---|---|---|---|---|"]
js = {}
for n in names:
    fv = out['FETCH_SIZE'][n]; wv = out['WRITE_SIZE'].get(n, [(0, 0)])
    fr = sum(v for v, _ in fv) / len(fv) / 1024; wr = sum(v for v, _ in wv) / len(wv) / 1024; us = sum(t for _, t in fv) / len(fv)
    L.append("| `%s` | %d | %.1f | %.1f | %.1f | %.1f | %.1f |" % (n[:48], len(fv), fr, 2 * fr, wr, 2 * fr + wr, us))
    js[n] = {"fetch_raw_MB": round(fr, 2), "read_MB": round(2 * fr, 2), "write_MB": round(wr, 2), "traffic_bytes_per_launch": int((2 * fr + wr) * 1048576), "avg_us": round(us, 1)}
tot = sum(v["traffic_bytes_per_launch"] for v in js.values())
L += ["", "Whole step: %.2f GB of HBM-side traffic per 256-frame batch = %.1f MB per frame, against B_alg = 195 MB per frame of the" % (tot / 1e9, tot / 256 / 1e6),
      "stream-once model (SURVEY.md section 8d)."]
open(f"{P}/{tag}_pmc.md", 'w').write("\n".join(L) + "\n")
json.dump({"tag": tag, "config": {"batch": 256, "geom": "64x2048", "clusters": 100}, "launches_per_step": 1, "kernels": js,
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, serial steps; read = 2 x FETCH_SIZE (gfx950, calibrated)"},
          open(f"{P}/r01_pmc.json", 'w'), indent=1)
print(open(f"{P}/{tag}_kernel_stats_serial.md").read())
print("FPS traffic MB:", [v for k, v in js.items() if k.startswith("fps_tiled_kernel<true")][0]["traffic_bytes_per_launch"] / 1e6, "total GB:", tot / 1e9)
print("pipelined fps avg us:", [v for k, v in d.items() if k.startswith("fps_tiled_kernel<true")], " bench launch_ms:", json.loads(open(f"{P}/{tag}_p_default_bench.json").read())["roofline"]["launch_ms"])

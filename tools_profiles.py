#!/usr/bin/env python3
"""Build a committed profiles/ set from the raw rocprofv3 CSVs that tools_dev/round_profiles.sh left in gpurun_out/ (scratch).
usage: tools_profiles.py <tag> [--config N] [--input] [--no-current] [--src DIR]      e.g. r03_v1 --src gpurun_out/c1
Writes profiles/<tag>_{kernel_stats_serial,kernel_stats_pipelined}.md (+ _raw.csv), <tag>_pmc.md, the three bench lines,
and (unless --no-current) profiles/pmc_current.json (pmc_current_c2.json for --config 2, ..._real.json for --input), which
bench.py reads for roofline.traffic / valu_wave_insts when it runs that configuration."""
import collections, csv, json, os, sys
tag = sys.argv[1]
cfg = int(sys.argv[sys.argv.index("--config") + 1]) if "--config" in sys.argv else 1
real = "--input" in sys.argv
G = sys.argv[sys.argv.index("--src") + 1] if "--src" in sys.argv else "gpurun_out"
P = "profiles"
SKIP = ('at::native', 'rocblas', 'rocclr', 'rocprim', 'hipcub', 'anonymous')
short = lambda n: n.split('(')[0].replace('void ', '')

def stats(src, dst_md, dst_csv):
    rows = list(csv.reader(open(src)))
    keep = [rows[0]] + [r for r in rows[1:] if not any(t in r[0] for t in SKIP)]
    csv.writer(open(dst_csv, 'w')).writerows(keep)
    L = ["| kernel | calls | avg us | min us | max us |", "|---|---|---|---|---|"]
    tot = 0.0
    for r in csv.DictReader(open(dst_csv)):
        L.append("| `%s` | %s | %.1f | %.1f | %.1f |" % (short(r['Name'])[:60], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
        tot += float(r['AverageNs']) / 1e3
    L.append("\nsum of the averages: %.1f us per batch (one launch of each per batch)" % tot)
    open(dst_md, 'w').write("\n".join(L) + "\n")
    return {short(r['Name']): float(r['AverageNs']) / 1e3 for r in csv.DictReader(open(dst_csv))}

d = stats(f"{G}/prof_default_kernel_stats.csv", f"{P}/{tag}_kernel_stats_pipelined.md", f"{P}/{tag}_kernel_stats_pipelined_raw.csv")
s = stats(f"{G}/prof_serial_kernel_stats.csv", f"{P}/{tag}_kernel_stats_serial.md", f"{P}/{tag}_kernel_stats_serial_raw.csv")
bench = {}
for n in ("prof_default_bench", "prof_serial_bench", "prof_bench_final"):
    line = open(f"{G}/{n}.json").read().strip().splitlines()[-1]
    open(f"{P}/{tag}_{n[5:]}.json", 'w').write(line + "\n")
    bench[n] = json.loads(line)

# static VALU cycle mix per kernel (tools_dev/isa_count.py --json): mean SIMD cycles per wave64 instruction by the measured classes
mix = json.load(open(f"{P}/isa_mix_current.json"))
def mix_of(kernel):
    if kernel not in mix:      # (no silent 4.0, no look-alike kernel: regenerate with `python tools_dev/isa_count.py --json profiles/isa_mix_current.json`)
        raise SystemExit("kernel %r is not in %s/isa_mix_current.json: regenerate it for this build" % (kernel, P))
    return mix[kernel]
def mean_cycles(kernel):
    return mix_of(kernel)["valu_mean_cycles"]

# dynamic class mix: SQ_INSTS_VALU_<class> per kernel launch (two PMC passes), weighted with the mean cycles of the kernel's STATIC
# instructions of that class (a class holds 2- and 4-cycle forms: plain / packed / DPP); what no class counter sees -- compares, selects,
# fp32 min / max, bit operations, moves, lane operations: SQ_INSTS_VALU minus the class sum -- keeps its static mean.  lo / hi: every
# class at the least / largest cycles its static instructions have.
CLASSES = ("ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "INT32", "INT64", "CVT", "ADD_F64", "MUL_F64", "FMA_F64", "TRANS_F64")
dyn = {}
if all(os.path.exists(f"{G}/prof_pmc_CLASS_{x}.csv") for x in "AB"):
    raw = collections.defaultdict(lambda: collections.defaultdict(list))
    for x in "AB":
        for r in csv.DictReader(open(f"{G}/prof_pmc_CLASS_{x}.csv")):
            if not any(t in r['Kernel_Name'] for t in SKIP):
                raw[short(r['Kernel_Name'])][r['Counter_Name'] + ("@" + x if r['Counter_Name'] == "SQ_INSTS_VALU" else "")].append(float(r['Counter_Value']))
    for k, v in raw.items():
        avg = {c: sum(x) / len(x) for c, x in v.items()}
        total = avg.get("SQ_INSTS_VALU@A") or avg.get("SQ_INSTS_VALU@B")
        if not total:      # (a launch that leaves at once -- the whole-cloud kernels of a batch without ground-less sweeps -- counts nothing: static mix)
            continue
        cl = mix_of(k)["pmc_classes"]
        n = {c: avg.get("SQ_INSTS_VALU_" + c, 0.0) for c in CLASSES}
        n["UNCOUNTED"] = max(total - sum(n.values()), 0.0)
        cyc = lo = hi = 0.0
        for c, cnt in n.items():
            st = cl.get(c) or {"cycles": 4.0, "lo": 2, "hi": 4}    # (a class the static code does not hold but the counter reports: rare aliasing)
            cyc += cnt * st["cycles"]; lo += cnt * st["lo"]; hi += cnt * st["hi"]
        dyn[k] = {"insts": total, "cycles": cyc / total, "lo": lo / total, "hi": hi / total, "share": {c: round(cnt / total, 4) for c, cnt in n.items() if cnt},
                  "static_share": {c: round(st["n"] / mix_of(k)["valu_static"], 4) for c, st in cl.items()}}

out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f"{G}/prof_pmc_{c}.csv")):
        n = r['Kernel_Name']
        if any(t in n for t in SKIP) or r['Counter_Name'] != c:
            continue
        agg[short(n)].append((float(r['Counter_Value']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
    out[c] = agg
names = sorted(out['FETCH_SIZE'], key=lambda n: -sum(v for v, _ in out['FETCH_SIZE'][n]))
wl = bench["prof_bench_final"]["config"]["workload"]
L = [f"# {tag} PMC passes: HBM-side traffic and VALU instructions per kernel launch (serial steps)", "", "workload: " + wl, "",
     "Three separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md; no trace domain besides --kernel-trace):", "",
     "    rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU> --kernel-trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-verify --pipeline 1 [config flags]", "",
     "Units: the size counters are KB.  gfx950 correction (guide, section HBM): FETCH_SIZE reports half of the bytes of a coalesced",
     "stream.  Calibration on kernels with known byte counts: `project_pix_kernel` reads 12 B x 29.0 M points = 348 MB,",
     "`assign_kernel` reads the 134 MB range image; both report about half, so read = 2 x FETCH_SIZE also for 4 B/lane loads.",
     "WRITE_SIZE needs no correction (`project_pix_kernel` writes 232 MB of records, `assign_kernel` 33.5 MB of labels).",
     "SQ_INSTS_VALU counts wave-level VALU instructions.  A wave64 instruction occupies its SIMD for 2, 4 or 8 cycles depending on its class (measured: profiles/r04_valu_peak.md);",
     "mean cycles = the launch's dynamic class counts (below) weighted with the static cycles inside each class -- the kernel's static mix alone when the class passes are missing --",
     "(profiles/isa_mix_current.json); VALU busy = instructions x mean cycles / (1024 SIMDs x 2.4 GHz x duration).", "",
     "| kernel | launches | FETCH_SIZE raw MB | read MB (x2) | WRITE_SIZE MB | HBM-side traffic MB/launch | VALU M wave-instr./launch | mean cycles / instr. | avg us (profiled) | traffic TB/s | VALU busy % |", "|---|---|---|---|---|---|---|---|---|---|---|"]
js = {}
for n in names:
    fv = out['FETCH_SIZE'][n]; wv = out['WRITE_SIZE'].get(n, [(0, 0)]); vv = out['SQ_INSTS_VALU'].get(n, [(0, 0)])
    fr = sum(v for v, _ in fv) / len(fv) / 1024; wr = sum(v for v, _ in wv) / len(wv) / 1024; us = sum(t for _, t in fv) / len(fv)
    va = sum(v for v, _ in vv) / len(vv)
    tb = (2 * fr + wr) * 1048576 / (us * 1e-6) / 1e12 if us else 0
    mc = dyn[n]["cycles"] if n in dyn else mean_cycles(n)
    vp = va * mc / (us * 1e-6) / (1024 * 2.4e9) * 100 if us else 0
    L.append("| `%s` | %d | %.1f | %.1f | %.1f | %.1f | %.1f | %.2f | %.1f | %.2f | %.0f |" % (n[:48], len(fv), fr, 2 * fr, wr, 2 * fr + wr, va / 1e6, mc, us, tb, vp))
    js[n] = {"fetch_raw_MB": round(fr, 2), "read_MB": round(2 * fr, 2), "write_MB": round(wr, 2), "traffic_bytes_per_launch": int((2 * fr + wr) * 1048576),
             "valu_wave_insts_per_launch": int(va), "valu_mean_cycles_static": mean_cycles(n), "avg_us": round(us, 1)}
    if n in dyn:
        js[n].update(valu_mean_cycles_dynamic=round(dyn[n]["cycles"], 4), valu_mean_cycles_lo=round(dyn[n]["lo"], 4), valu_mean_cycles_hi=round(dyn[n]["hi"], 4))
tot = sum(v["traffic_bytes_per_launch"] for v in js.values()); totv = sum(v["valu_wave_insts_per_launch"] for v in js.values())
best = lambda v: v.get("valu_mean_cycles_dynamic", v["valu_mean_cycles_static"])
totc = sum(v["valu_wave_insts_per_launch"] * best(v) for v in js.values())
totc_static = sum(v["valu_wave_insts_per_launch"] * v["valu_mean_cycles_static"] for v in js.values())
totc_lo = sum(v["valu_wave_insts_per_launch"] * v.get("valu_mean_cycles_lo", 2.0) for v in js.values())
totc_hi = sum(v["valu_wave_insts_per_launch"] * v.get("valu_mean_cycles_hi", 4.0) for v in js.values())
B = bench["prof_bench_final"]["config"]["frames_per_gpu_per_step"]
L += ["", "Whole step: %.2f GB of HBM-side traffic per %d-frame batch = %.1f MB per frame, %d launches, %.0f M wave-level VALU instructions" % (tot / 1e9, B, tot / B / 1e6, len(js), totv / 1e6),
      "x %.2f cycles (mix-weighted mean) = %.3f ms of VALU time on the whole chip (1024 SIMDs at 2.4 GHz)." % (totc / max(totv, 1), totc / (1024 * 2.4e9) * 1e3)]
if dyn:
    L += ["", "mean cycles / instr.: the DYNAMIC class counts of the launch (SQ_INSTS_VALU_ADD_F32 / MUL_F32 / FMA_F32 / TRANS_F32 / INT32 / INT64 / CVT and the fp64 four, two",
          "passes) weighted with the mean cycles of the kernel's static instructions of each class; instructions no class counter sees (compares, selects,",
          "fp32 min / max, bit operations, moves, lane operations) keep their static mean.  Static mix alone: %.3f cycles; bounds with every class at its" % (totc_static / max(totv, 1)),
          "least / largest static cycles: %.3f .. %.3f." % (totc_lo / max(totv, 1), totc_hi / max(totv, 1)), "",
          "| kernel | static mean | dynamic mean | lo .. hi | dynamic share of: f32 add+mul+fma | trans | int32 | cvt | fp64 | uncounted | static share of uncounted |", "|---|---|---|---|---|---|---|---|---|---|---|"]
    for n in names:
        if n not in dyn:
            continue
        d_, sh = dyn[n], dyn[n]["share"]
        f64 = sum(sh.get(c, 0) for c in ("ADD_F64", "MUL_F64", "FMA_F64", "TRANS_F64"))
        L.append("| `%s` | %.2f | %.2f | %.2f .. %.2f | %.2f | %.3f | %.2f | %.3f | %.3f | %.2f | %.2f |" % (
            n[:48], mean_cycles(n), d_["cycles"], d_["lo"], d_["hi"], sh.get("ADD_F32", 0) + sh.get("MUL_F32", 0) + sh.get("FMA_F32", 0), sh.get("TRANS_F32", 0),
            sh.get("INT32", 0) + sh.get("INT64", 0), sh.get("CVT", 0), f64, sh.get("UNCOUNTED", 0), d_["static_share"].get("UNCOUNTED", 0)))
open(f"{P}/{tag}_pmc.md", 'w').write("\n".join(L) + "\n")
if "--no-current" not in sys.argv:
    geom = wl.split("(")[-1].split(")")[0] if "x" in wl else "64x2048"
    import re
    m = re.search(r"\((\d+x\d+)\)", wl)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import rpcc_amd  # noqa: F401
    from rpcc_amd.build import source_digest
    # the counters describe the kernels of THIS source tree: run this script on the tree the passes were taken with (bench.py checks)
    json.dump({"tag": tag, "source_sha256": source_digest(), "config": {"batch": B, "geom": m.group(1) if m else "64x2048", "clusters": 100, "config": cfg, "input": real},
               "step_traffic_bytes": tot, "step_valu_wave_insts": totv, "step_valu_simd_cycles": int(totc), "step_valu_simd_cycles_static_mix": int(totc_static),
               "step_valu_simd_cycles_lo": int(totc_lo), "step_valu_simd_cycles_hi": int(totc_hi), "valu_cycles_source": "dynamic class counters" if dyn else "static mix", "kernels": js,
               "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU, separate passes, serial steps; read = 2 x FETCH_SIZE (gfx950, calibrated); "
                       "valu_mean_cycles_static: static instruction mix of the kernel's assembly weighted with the measured cycle classes (profiles/r04_valu_peak.md); "
                       "valu_mean_cycles_dynamic: SQ_INSTS_VALU_<class> counts of the launch x the static mean cycles inside each class (lo / hi: every class at its least / largest cycles)"},
              open(f"{P}/pmc_current.json" if cfg == 1 and not real else f"{P}/pmc_current_c{cfg}{'_real' if real else ''}.json", 'w'), indent=1)
print(open(f"{P}/{tag}_kernel_stats_serial.md").read())
print(open(f"{P}/{tag}_pmc.md").read().split("| kernel |")[1][:4000])
print("bench (no profiler):", bench["prof_bench_final"]["value"], "frames/s", bench["prof_bench_final"]["ms_per_step"], "ms/step; serial:", bench["prof_serial_bench"]["ms_per_step"])

#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats CSV: keep this library's kernels, drop torch's
(synthetic-data generator) -- usage: tools_prof_summary.py <dir> [out.md]"""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + '/**/*kernel_stats.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if not any(t in r['Name'] for t in ('at::native', 'rocblas', 'rocclr', 'rocprim', 'hipcub'))]
lines = ["| kernel | calls | avg us | min us | max us |", "|---|---|---|---|---|"]
for r in rows:
    lines.append("| `%s` | %s | %.1f | %.1f | %.1f |" % (r['Name'].split('(')[0][:60], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], 'w').write(out + "\n")

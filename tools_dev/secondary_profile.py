#!/usr/bin/env python3
"""Build profiles/r01_secondary_kernel_stats.md from gpurun_out/secondary_kernel_stats.csv (rocprofv3 --kernel-trace --stats
of tools_dev/secondary_bench.py).  usage: secondary_profile.py "<wall-clock sentence>" """
import csv, sys
SKIP = ('at::native', 'rocclr', 'rocblas', 'rocprim', 'hipcub', 'anonymous')
rows = list(csv.DictReader(open('gpurun_out/secondary_kernel_stats.csv')))
keep = [r for r in rows if not any(t in r['Name'] for t in SKIP)]
w = csv.DictWriter(open('profiles/r01_secondary_kernel_stats_raw.csv', 'w'), fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(keep)
L = ["# r01 secondary rows: kernel statistics of `tools_dev/secondary_bench.py` (B = 256 frames of 64x2048)", "",
     "    rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools_dev/secondary_bench.py", "",
     "The script runs the four framework / model combinations through `BatchCompressor.compress_device` (6 calls each), then",
     "contour encode / decode, payload packing and the decoder alone.  Wall clock of the same run without the profiler:",
     sys.argv[1],
     "Torch's own kernels (the synthetic-frame generator of the set-up and the buffer copies of the stage-by-stage path) are",
     "filtered out of the table.", "",
     "| kernel | calls | avg us | min us | max us |", "|---|---|---|---|---|"]
for r in keep:
    L.append("| `%s` | %s | %.1f | %.1f | %.1f |" % (r['Name'].split('(')[0].replace('void ', '')[:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
open('profiles/r01_secondary_kernel_stats.md', 'w').write("\n".join(L) + "\n")
print("\n".join(L))

#!/bin/bash
# GPU box: everything the committed profiles/ set is made from (then: python tools_profiles.py <tag> locally)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf /tmp/rp && mkdir -p /tmp/rp
python3 bench.py > gpurun_out/bench_final.json 2> /tmp/rp/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp/d -o p -- python3 bench.py > gpurun_out/p_default_bench.json 2>/tmp/rp/e1
cp $(find /tmp/rp/d -name "*kernel_stats.csv" | head -1) gpurun_out/p_default_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp/s -o p -- python3 bench.py --pipeline 1 > gpurun_out/p_serial_bench.json 2>/tmp/rp/e2
cp $(find /tmp/rp/s -name "*kernel_stats.csv" | head -1) gpurun_out/p_serial_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/rp/$c -o p -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --pipeline 1 > /tmp/rp/$c.log 2>&1
  python3 - $(find /tmp/rp/$c -name "*counter_collection.csv" | head -1) gpurun_out/pmc2_$c.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if not any(t in r['Kernel_Name'] for t in ('at::native', 'rocprim', 'hipcub', 'rocblas', 'rocclr'))]
w = csv.DictWriter(open(sys.argv[2], 'w'), fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(keep)
PY
done
tail -c 600 gpurun_out/bench_final.json

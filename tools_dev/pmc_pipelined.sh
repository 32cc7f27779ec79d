#!/bin/bash
# developer helper (GPU box): one PMC pass over a short PIPELINED bench run (three batches in flight); per-kernel averages
# usage: tools_dev/pmc_pipelined.sh "CTR1 CTR2 ..." [pipeline depth]
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
rm -rf /tmp/pmcp && mkdir -p /tmp/pmcp
rocprofv3 --pmc $1 --kernel-trace --output-format csv -d /tmp/pmcp -o p -- python3 bench.py --no-secondary --cpu-sample 0 --no-verify --pipeline ${2:-3} --steps 6 --warmup 3 > /tmp/pmcp/bench.log 2>&1
f=$(find /tmp/pmcp -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    if any(t in n for t in ('at::native', 'rocprim', 'hipcub', 'rocblas', 'rocclr')): continue
    agg[n.split('(')[0].replace('void ', '')[:40]][r['Counter_Name']].append(float(r['Counter_Value']))
ctrs = sorted({c for v in agg.values() for c in v})
print("%-42s" % "kernel" + "".join("%18s" % c[:17] for c in ctrs))
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1][ctrs[0]]) / max(len(kv[1][ctrs[0]]), 1)):
    print("%-42s" % n + "".join("%18.4g" % (sum(v[c]) / max(len(v[c]), 1)) for c in ctrs))
PY
tail -1 /tmp/pmcp/bench.log | cut -c1-200

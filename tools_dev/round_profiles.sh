#!/bin/bash
# GPU box: everything a committed profiles/ set is made from (then: python tools_profiles.py <tag> locally).
# usage: tools_dev/round_profiles.sh [extra bench.py args, e.g. --config 2]      -> gpurun_out/prof_*
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf /tmp/rp && mkdir -p /tmp/rp
timeout 900 python3 bench.py "$@" > gpurun_out/prof_bench_final.json 2> /tmp/rp/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp/d -o p -- python3 bench.py --no-secondary "$@" > gpurun_out/prof_default_bench.json 2>/tmp/rp/e1
cp $(find /tmp/rp/d -name "*kernel_stats.csv" | head -1) gpurun_out/prof_default_kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp/s -o p -- python3 bench.py --no-secondary --pipeline 1 "$@" > gpurun_out/prof_serial_bench.json 2>/tmp/rp/e2
cp $(find /tmp/rp/s -name "*kernel_stats.csv" | head -1) gpurun_out/prof_serial_kernel_stats.csv
# PMC passes, each on its own (FETCH_SIZE and WRITE_SIZE do not fit one pass; no trace domains besides --kernel-trace)
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/rp/$c -o p -- python3 bench.py --no-secondary --steps 3 --warmup 1 --cpu-sample 0 --no-verify --pipeline 1 "$@" > /tmp/rp/$c.log 2>&1
  python3 - $(find /tmp/rp/$c -name "*counter_collection.csv" | head -1) gpurun_out/prof_pmc_$c.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if not any(t in r['Kernel_Name'] for t in ('at::native', 'rocprim', 'hipcub', 'rocblas', 'rocclr'))]
w = csv.DictWriter(open(sys.argv[2], 'w'), fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(keep)
PY
done
# the dynamic VALU class mix (SQ_INSTS_VALU_*: two passes of eight counters; tools_profiles.py weights them with the kernel's static cycles per class)
PA="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT"
PB="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64"
for pass in A B; do
  [ $pass == A ] && PC="$PA" || PC="$PB"
  timeout 600 rocprofv3 --pmc $PC --kernel-trace --output-format csv -d /tmp/rp/CLASS_$pass -o p -- python3 bench.py --no-secondary --steps 3 --warmup 1 --cpu-sample 0 --no-verify --pipeline 1 "$@" > /tmp/rp/CLASS_$pass.log 2>&1
  python3 - $(find /tmp/rp/CLASS_$pass -name "*counter_collection.csv" | head -1) gpurun_out/prof_pmc_CLASS_$pass.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if not any(t in r['Kernel_Name'] for t in ('at::native', 'rocprim', 'hipcub', 'rocblas', 'rocclr'))]
w = csv.DictWriter(open(sys.argv[2], 'w'), fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(keep)
PY
done
tail -c 300 gpurun_out/prof_bench_final.json

#!/bin/bash
# GPU box: the dynamic VALU instruction mix of the headline's kernels by the SQ_INSTS_VALU_* class counters, and what those counters
# count (the microbenchmark's one-instruction kernels under the same counters)  ->  profiles/r04_valu_peak.md
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out
[ -x /tmp/valu_peak ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools_dev/valu_peak.hip -o /tmp/valu_peak 2>/dev/null || exit 1
PA="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT"
PB="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_IOPS SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_FLOPS_FP64"
for pass in A B; do
  [ $pass == A ] && P="$PA" || P="$PB"
  rm -rf /tmp/vm$pass && mkdir -p /tmp/vm$pass
  timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/vm$pass -o p -- /tmp/valu_peak 300 > /tmp/vm$pass/run.log 2>&1
  python3 - $(find /tmp/vm$pass -name "*counter_collection.csv" | head -1) > gpurun_out/valu_mix_cal_$pass.txt <<'PY'
import csv, sys, collections, re
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"peak_kernel<(\d+), (\d+)>", r['Kernel_Name'])
    if not m or int(m.group(2)) != 8 or int(r['Grid_Size']) // 64 // 1024 != 8: continue
    agg.setdefault(int(m.group(1)), {})[r['Counter_Name']] = float(r['Counter_Value'])
ctrs = sorted({c for v in agg.values() for c in v} - {'SQ_INSTS_VALU'})
print("| op | " + " | ".join(c.replace('SQ_INSTS_VALU_', '') for c in ctrs) + " |   (counter / SQ_INSTS_VALU of a kernel made of that one instruction)")
print("|---|" + "---|" * len(ctrs))
for op, v in agg.items():
    iv = max(v.get('SQ_INSTS_VALU', 1), 1)
    print("| %d | " % op + " | ".join("%.2f" % (v.get(c, 0) / iv) for c in ctrs) + " |")
PY
  tools_dev/pmc.sh "$P" > gpurun_out/valu_mix_$pass.txt 2>&1
done

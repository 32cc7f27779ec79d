#!/bin/bash
# GPU box: the VALU issue-rate microbenchmark, the same kernels under the SQ counters (what SQ_ACTIVE_INST_VALU counts per
# instruction class), and the SQ busy / active counters of the headline's kernels  ->  profiles/r04_valu_peak.md
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools_dev/valu_peak.hip -o /tmp/valu_peak 2>/dev/null || exit 1
timeout 600 /tmp/valu_peak > gpurun_out/valu_peak.txt 2>&1
rm -rf /tmp/vp && mkdir -p /tmp/vp
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/vp -o p -- /tmp/valu_peak 4000 > /tmp/vp/run.log 2>&1
python3 - $(find /tmp/vp -name "*counter_collection.csv" | head -1) > gpurun_out/valu_peak_pmc.txt <<'PY'
import csv, sys, collections, re
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"peak_kernel<(\d+), (\d+)>", r['Kernel_Name'])
    if not m: continue
    key = (int(m.group(1)), int(m.group(2)), int(r['Grid_Size']) // 64 // 1024)   # op, chain, wavefronts per SIMD
    agg.setdefault(key, collections.defaultdict(list))[r['Counter_Name']].append(float(r['Counter_Value']))
print("| op | chain | W | SQ_INSTS_VALU | ACTIVE_INST_VALU / INSTS_VALU | THREAD_CYCLES_VALU / INSTS_VALU | ACTIVE_INST_VALU x 4 / (1024 x GRBM_GUI_ACTIVE / 8) | INSTS_SALU | ACTIVE_INST_SCA / INSTS_SALU |")
print("|---|---|---|---|---|---|---|---|---|")
for (op, ch, w), v in agg.items():
    last = {c: x[-1] for c, x in v.items()}      # the last (full-length) launch of the configuration
    iv = max(last.get('SQ_INSTS_VALU', 0), 1); isa = max(last.get('SQ_INSTS_SALU', 0), 1)
    print("| %d | %d | %d | %.4g | %.3f | %.1f | %.3f | %.4g | %.3f |" % (op, ch, w, iv, last.get('SQ_ACTIVE_INST_VALU', 0) / iv, last.get('SQ_THREAD_CYCLES_VALU', 0) / iv,
          last.get('SQ_ACTIVE_INST_VALU', 0) * 4 / max(1024 * last.get('GRBM_GUI_ACTIVE', 1) / 8, 1), isa, last.get('SQ_ACTIVE_INST_SCA', 0) / isa))
PY
tools_dev/pmc.sh "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" > gpurun_out/pmc_sq1.txt 2>&1
tools_dev/pmc.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU" > gpurun_out/pmc_sq2.txt 2>&1

"""Registers, scratch and occupancy of every kernel of librpcc_hip.so as the compiler reports them (no GPU needed).
usage: python tools_dev/regs.py [substring ...]   (extra hipcc flags through RPCC_EXTRA_FLAGS, as for the build)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rpcc_amd  # noqa: F401
from rpcc_amd import build as b
cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + b.HIPCC_FLAGS + os.environ.get("RPCC_EXTRA_FLAGS", "").split() + \
      ["-Rpass-analysis=kernel-resource-usage", os.environ.get("RPCC_SRC", b.SRC), "-o", "/tmp/rpcc_regs.so"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|VGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().split("(")[0]}
        rows.append(cur)
    elif cur is not None:
        cur[k.split(" ")[0]] = v
pats = sys.argv[1:]
print("%-72s %5s %5s %7s %4s %6s" % ("kernel", "VGPR", "SGPR", "scratch", "occ", "LDS"))
for r in rows:
    if pats and not any(p in r["name"] for p in pats):
        continue
    print("%-72s %5s %5s %7s %4s %6s" % (r["name"][-72:], r.get("VGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize"), r.get("Occupancy"), r.get("LDS")))

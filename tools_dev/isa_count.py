#!/usr/bin/env python3
"""Static instruction counts per kernel from the gfx950 assembly of librpcc_hip.so (no GPU needed).

    python tools_dev/isa_count.py [--md profiles/rNN_isa_counts.md] [--dump KERNEL_SUBSTRING] [extra hipcc flags]

Compiles r-pcc_amd/csrc/rpcc_hip.hip with the product flags + --save-temps into a scratch directory and, per kernel,
counts the instructions by class (VALU split into fp64 / packed / transcendental / DPP / other), the register and LDS
budget from the kernel descriptor, and scratch use.  Static counts are not dynamic counts (loops, branches), but for the
straight-line throughput kernels of the path (pixel, mask, assign's per-tile body, quantiser) they are what the
SQ_INSTS_VALU counter multiplies by the number of wavefronts.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def compile_asm(extra):
    import rpcc_amd  # noqa: F401
    from rpcc_amd import build as b
    d = tempfile.mkdtemp(prefix="rpcc_isa_")
    cmd = ["/opt/rocm/bin/hipcc"] + b.HIPCC_FLAGS + extra + ["--save-temps", b.SRC, "-o", os.path.join(d, "lib.so")]
    subprocess.check_call(cmd, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return [os.path.join(d, f) for f in os.listdir(d) if f.endswith("gfx950.s")][0]


def demangle(names):
    out = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
    return {n: re.sub(r"^void ", "", o.split("(")[0]) for n, o in zip(names, out)}


def classify(op):
    if op.startswith("v_"):
        if "_f64" in op:
            return "valu_f64"
        if op.startswith("v_pk_"):
            return "valu_pk"
        if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_", op):
            return "valu_trans"
        if op.endswith("_dpp") or "_dpp" in op:
            return "valu_dpp"
        if re.match(r"v_(readlane|readfirstlane|writelane)", op):
            return "valu_lane"
        if op.startswith("v_cmp") or op.startswith("v_cmpx"):
            return "valu_cmp"
        if op.startswith("v_cndmask"):
            return "valu_sel"
        if op.startswith("v_div_"):
            return "valu_div"
        return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_load") or op.startswith("flat_load") or op.startswith("buffer_load"):
        return "vmem_ld"
    if op.startswith("global_store") or op.startswith("flat_store") or op.startswith("buffer_store"):
        return "vmem_st"
    if op.startswith("global_atomic") or op.startswith("flat_atomic") or op.startswith("buffer_atomic"):
        return "vmem_atomic"
    if op.startswith("scratch_"):
        return "scratch"
    return "other"


# SIMD cycles a wave64 instruction occupies the VALU for (measured: tools_dev/valu_peak.hip, profiles/r04_valu_peak.md).
FAST2 = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_fmac_f32", "v_mac_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
         "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32"}


def valu_cycles(op):
    base = re.sub(r"_(e32|e64|sdwa)$", "", op)
    if "_dpp" in base:
        return 4
    if re.match(r"v_(rcp|rsq|sqrt)_f64", base):
        return 16
    if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_", base):
        return 8
    return 2 if base in FAST2 else 4


# Which SQ_INSTS_VALU_* class counter sees an instruction (calibrated with one-instruction kernels: profiles/raw/r04_valu_mix_cal_A.txt, _B):
# the counters count plain, packed and DPP forms alike; compares, selects, fp32 min / max / med3, bit operations, left / logical right shifts,
# moves, lane operations, roundings count in NONE of them ("uncounted").  tools_profiles.py weights the DYNAMIC class counts of a kernel with
# the mean cycles of the kernel's static instructions of that class.
PMC_CLASS_RULES = (
    ("TRANS_F64", r"v_(rcp|rsq|sqrt)_f64"), ("FMA_F64", r"v_fma_f64"), ("MUL_F64", r"v_mul_f64"), ("ADD_F64", r"v_add_f64"),
    ("TRANS_F32", r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_(f32|legacy_f32|iflag_f32)"),
    ("FMA_F32", r"v_(pk_)?(fma|fmac|mad|mac)_f32"), ("MUL_F32", r"v_(pk_)?mul_(legacy_)?f32"), ("ADD_F32", r"v_(pk_)?(add|sub|subrev)_f32"),
    ("CVT", r"v_cvt_"), ("INT64", r"v_mad_[ui]64_[ui]32"),
    ("INT32", r"v_(add|sub|subrev|addc|subb|subbrev)(_co)?_[ui]32|v_(lshl_add|add_lshl|add3|xad)_u32|v_mbcnt_|v_mul_(lo|hi)_[ui]32|v_mul_[ui]32_[ui]24|"
              r"v_mad_[ui]32_[ui]24|v_ashrrev_i32|v_(min|max)_[ui]32|v_bfe_[ui]32|v_ffb[hl]_"),
)


def pmc_class(op):
    base = re.sub(r"_(e32|e64|sdwa|dpp|e64_dpp)$", "", op)
    for name, rx in PMC_CLASS_RULES:
        if re.match(rx, base):
            return name
    return "UNCOUNTED"


def parse(path):
    kernels, cur, body = {}, None, []
    meta = collections.defaultdict(dict)
    for ln in open(path):
        m = re.match(r"^([A-Za-z_][\w$.]*):\s*(;.*)?$", ln)
        if m and not m.group(1).startswith(".L") and not m.group(1).startswith("BB"):
            cur = m.group(1)
            body = kernels.setdefault(cur, [])
            continue
        s = ln.strip()
        if s.startswith(".amdhsa_") and cur:
            k, _, v = s.partition(" ")
            meta[cur][k] = v.strip()
        if s.startswith("; ") and cur:
            m2 = re.match(r"; (NumVgprs|NumAgprs|NumSgprs|ScratchSize|LDSByteSize|Occupancy): (\d+)", s)
            if m2:
                meta[cur][m2.group(1)] = int(m2.group(2))
        if cur and s and not s.startswith((".", ";", "BB", "//")) and not s.endswith(":"):
            body.append(s.split()[0])
        if s.startswith(".end_amdhsa_kernel") or s.startswith(".Lfunc_end"):
            pass
    return kernels, meta


def main():
    args = sys.argv[1:]
    md = dump = None
    if "--md" in args:
        i = args.index("--md"); md = args[i + 1]; del args[i:i + 2]
    js = None
    if "--json" in args:
        i = args.index("--json"); js = args[i + 1]; del args[i:i + 2]
    if "--dump" in args:
        i = args.index("--dump"); dump = args[i + 1]; del args[i:i + 2]
    asm = compile_asm(args)
    kernels, meta = parse(asm)
    names = [k for k in kernels if "NumVgprs" in meta.get(k, {}) and k.startswith("_Z")]
    dm = demangle(names)
    classes = ["valu", "valu_f64", "valu_pk", "valu_trans", "valu_div", "valu_dpp", "valu_lane", "valu_cmp", "valu_sel", "salu", "smem", "lds",
               "vmem_ld", "vmem_st", "vmem_atomic", "scratch", "branch", "barrier", "wait"]
    rows = []
    for k in names:
        c = collections.Counter(classify(op) for op in kernels[k])
        tot_valu = sum(v for kk, v in c.items() if kk.startswith("valu"))
        vops = [op for op in kernels[k] if op.startswith("v_")]
        c["cyc"] = sum(valu_cycles(op) for op in vops) / max(len(vops), 1)
        c["fast2"] = sum(1 for op in vops if valu_cycles(op) == 2) / max(len(vops), 1)
        per = collections.defaultdict(list)
        for op in vops:
            per[pmc_class(op)].append(valu_cycles(op))
        # per class counter: static instruction count, their mean / least / largest cycles
        c["pmc"] = {k: {"n": len(v), "cycles": round(sum(v) / len(v), 4), "lo": min(v), "hi": max(v)} for k, v in sorted(per.items())}
        rows.append((dm[k], tot_valu, c, meta[k]))
        if dump and dump in dm[k]:
            print("==== " + dm[k])
            ops = collections.Counter(op for op in kernels[k] if op.startswith("v_"))
            for op, n in ops.most_common(60):
                print("  %-28s %d" % (op, n))
    rows.sort(key=lambda r: -r[1])
    hdr = "| kernel | VALU total | share of 2-cycle ops | mean cycles per VALU instr. | " + " | ".join(c.replace("valu_", "v:") for c in classes) + " | VGPR | AGPR | SGPR | scratch B | LDS B | occupancy |"
    L = [hdr, "|" + "---|" * (len(classes) + 10)]
    for n, tv, c, m in rows:
        L.append("| `%s` | %d | %.2f | %.2f | " % (n[:70], tv, c["fast2"], c["cyc"]) + " | ".join(str(c.get(cl, 0)) for cl in classes) +
                 " | %s | %s | %s | %s | %s | %s |" % (m.get("NumVgprs"), m.get("NumAgprs"), m.get("NumSgprs"), m.get("ScratchSize"), m.get("LDSByteSize"), m.get("Occupancy")))
    text = "\n".join(L) + "\n"
    if js:   # kernel -> static mean VALU cycles per instruction and share of 2-cycle operations (tools_profiles.py, bench.py's roofline)
        import json
        json.dump({n: {"valu_mean_cycles": round(c["cyc"], 4), "share_2cycle": round(c["fast2"], 4), "valu_static": tv, "pmc_classes": c["pmc"]} for n, tv, c, m in rows},
                  open(js, "w"), indent=1, sort_keys=True)
    if md:
        open(md, "w").write("# Static gfx950 instruction counts per kernel (hipcc --save-temps, product flags%s)\n\n" % ((" + " + " ".join(args)) if args else "") +
                            "`valu` = plain VALU; v:f64 fp64, v:pk packed fp32, v:trans rcp/sqrt/..., v:div the div_scale/fmas/fixup helpers of an IEEE "
                            "division, v:dpp DPP forms, v:lane readlane/readfirstlane, v:cmp compares, v:sel cndmask.  Static counts: loops and branches "
                            "are counted once.  Share of 2-cycle ops / mean cycles: every VALU instruction weighted with the SIMD cycles its class occupies "
                            "(2: fma / mul / add / sub f32, add / sub u32, and / or / xor / not, right shifts, mov; 8: fp32 transcendentals; 16: fp64 rcp / sqrt; 4: "
                            "everything else -- measured, profiles/r04_valu_peak.md).\n\n" + text)
    else:
        print(text)
    print("asm:", asm)


if __name__ == "__main__":
    main()

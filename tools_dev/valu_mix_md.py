"""Rebuild profiles/<round>_valu_mix_dynamic.md from a round's committed sets: the per-kernel tables of <tag>_pmc.md, <tag>_c2_pmc.md, <tag>_real_pmc.md
and the prices of profiles/pmc_current.json against the un-profiled bench line of the same session.   usage: python tools_dev/valu_mix_md.py r05_v3"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1]
rnd = tag.split("_")[0]


def tail(name):   # "Whole step: ..." paragraph + the static / dynamic table of a _pmc.md
    L = open(os.path.join(P, name)).read().splitlines()
    i = max(k for k, l in enumerate(L) if l.startswith("Whole step:"))
    j = max(k for k, l in enumerate(L) if l.startswith("| kernel | static mean"))
    whole = []
    k = i
    while L[k].strip():
        whole.append(L[k]); k += 1
    stat = [l for l in L[k:j] if l.startswith("Static mix alone") or "Static mix alone" in l]
    m = [l for l in L[k:j] if "Static mix alone" in l]
    line = ""
    if m:
        s = m[0]
        line = "Static mix alone: " + s.split("Static mix alone: ")[1].replace("; bounds with every class at its", " per instruction; class bounds").strip()
        nxt = L[L.index(s) + 1] if "least / largest" in L[L.index(s) + 1] else ""
        if nxt: line += " " + nxt.replace("least / largest static cycles:", "").strip()
    tab = []
    k = j
    while k < len(L) and L[k].startswith("|"):
        tab.append(L[k]); k += 1
    return "\n".join(whole) + ("\n" + line if line else "") + "\n\n" + "\n".join(tab) + "\n"


cur = json.load(open(os.path.join(P, "pmc_current.json")))
b = json.loads(open(os.path.join(P, tag + "_bench_final.json")).read())
ms = b["ms_per_step"]
peak = 1024 * 2.4e9 * ms * 1e-3
n = cur["step_valu_wave_insts"]
f = lambda c: "%.3f" % (c / peak)
intro = open(os.path.join(P, rnd + "_valu_mix_dynamic.md")).read().split("## configs[1]")[0]
out = intro
out += "## configs[1], the headline (256 x 64x2048, uniform + point model)\n\n" + tail(tag + "_pmc.md") + "\n"
out += "## configs[2] (256 x 64x2000, non-uniform + plane model)\n\n" + tail(tag + "_c2_pmc.md") + "\n"
out += "## the reference's real sweep replicated (256 x 64x2000)\n\n" + tail(tag + "_real_pmc.md") + "\n"
out += "## What it does to the headline's `roofline.frac`\n\n"
out += "Un-profiled run of the same session: %.4f ms per step (%.0f frames/s).  VALU pipe time / step time, 1024 SIMDs x 2.4 GHz:\n\n" % (ms, b["value"])
out += "| pricing | SIMD cycles per batch | frac |\n|---|---|---|\n"
out += "| dynamic class counts x static cycles inside the class (`frac`) | %.3e | %s |\n" % (cur["step_valu_simd_cycles"], f(cur["step_valu_simd_cycles"]))
out += "| static mix alone (rounds 3-4) | %.3e | %s |\n" % (cur["step_valu_simd_cycles_static_mix"], f(cur["step_valu_simd_cycles_static_mix"]))
out += "| every class at its least / largest static cycles | %.3e .. %.3e | %s .. %s |\n" % (
    cur["step_valu_simd_cycles_lo"], cur["step_valu_simd_cycles_hi"], f(cur["step_valu_simd_cycles_lo"]), f(cur["step_valu_simd_cycles_hi"]))
out += "| every instruction at 2 cycles (the guide's figure) / at 4 cycles (round 3) | %.3e / %.3e | %s / %s |\n\n" % (2 * n, 4 * n, f(2 * n), f(4 * n))
out += ("The dynamic weighting moves the step's mean from %.2f to %.2f cycles per instruction (mask and assign kernels: their fp64 fall-back sequences and the exact\n"
        "`atan2f` queue are cold; the FPS and histogram kernels are what their assembly says).  `bench.py` prints `frac`, `cycles_from` and the other prices as `frac_if`.\n"
        % (cur["step_valu_simd_cycles_static_mix"] / n, cur["step_valu_simd_cycles"] / n))
open(os.path.join(P, rnd + "_valu_mix_dynamic.md"), "w").write(out)
print("wrote", rnd + "_valu_mix_dynamic.md")

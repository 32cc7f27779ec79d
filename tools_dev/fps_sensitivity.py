#!/usr/bin/env python3
"""How much of the path's output depends on the two things of the reference's CUDA FPS binary that cannot be observed here
(DESIGN.md section 2 "parity unpinned"): nvcc's FMA contraction of sampling_gpu.cu:64 and the reduction tree's winner among
exactly equal distances.  CPU only (oracle/): for every frame the FPS runs in all six modes -- fma 0 / 1 / 2 x tie rule
lowest-index / CUDA tree -- on the candidates the reference path hands it, and the centre COORDINATES (what every later stage
consumes) are compared with the default mode's (un-fused, lowest index).

    python tools_dev/fps_sensitivity.py [--frames 256] [--md profiles/r03_fps_mode_sensitivity.md]

Frames: the bench batch (synthetic 64x2048, ids 0..N-1, ground plane by the seeded RANSAC as in bench.py), the reference's
example.bin (tests/golden/example_64E.npz, its golden ground model) and synthetic sweeps of the three shipped geometries.
"""
import argparse
import json
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MODES = [(f, t) for f in (0, 1, 2) for t in (False, True)]


def one(job):
    kind, arg = job
    import rpcc_amd  # noqa: F401
    from rpcc_amd import synth
    from oracle import oracle as orc
    if kind == "example":
        z = np.load(os.path.join(ROOT, "tests", "golden", "example_64E.npz"))
        g = orc.LidarGeom(**orc.GEOMS["Velodyne64E"])
        xyz, gm = z["xyz"], z["ground_model"]
        tm = orc.transform_map(g)
        ri = orc.project(xyz, g)
    else:
        name, fid = arg
        gd = dict(H=64, W=2048, hfov_deg=360, vmax_deg=2.0, vmin_deg=-24.9) if name == "bench" else orc.GEOMS[name]
        g = orc.LidarGeom(**gd)
        xyz = synth.make_frame(fid, g.H, g.W, vmax_deg=gd["vmax_deg"], vmin_deg=gd["vmin_deg"]).numpy()
        tm = orc.transform_map(g)
        ri = orc.project(xyz, g)
        gm = orc.ground_model(ri, tm, seed=fid)
    pc = orc.backproject(ri, tm)
    mask = orc.vertical_residual(pc, gm) > 0.1
    left = pc[np.where(mask)]
    base = None
    out = {}
    # exact ties at the arg-max between DISTINCT points, counted in the default arithmetic: what the tie rule can act on
    for f, t in MODES:
        idx = orc.fps_modes(left, 100, f, t)
        cen = left[idx]
        if base is None:
            base = cen
        # a centre "changes" when its coordinates differ as VALUES; the empty pixels all back-project to (+-0, +-0, +-0) -- the sign
        # of each zero follows the pixel's ray -- so two members of that class are the same point with different zero signs:
        # every distance computed from them is identical (x - (+0) == x - (-0), the squares are +0), only the stored bits differ
        dif = (cen != base).any(1)
        same = not dif.any()
        first = int(np.argmax(dif)) if not same else -1
        nchg = int(dif.sum())
        zsign = int(((cen.view(np.uint32) != base.view(np.uint32)).any(1) & ~dif).sum())    # same value, other zero signs
        out["%d%s" % (f, "c" if t else "l")] = (same, first, nchg, zsign)
    return kind if kind == "example" else arg[0], out, int(left.shape[0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--per-geom", type=int, default=16)
    ap.add_argument("--md", default=None)
    ap.add_argument("--workers", type=int, default=os.cpu_count() or 4)
    a = ap.parse_args()
    from oracle import oracle as orc
    orc.lib()
    jobs = [("synth", ("bench", i)) for i in range(a.frames)] + [("example", None)]
    for name in ("Velodyne64E", "Velodyne32E", "VelodyneVLP16"):
        jobs += [("synth", (name, 5000 + i)) for i in range(a.per_geom)]
    with ProcessPoolExecutor(a.workers) as ex:
        res = list(ex.map(one, jobs, chunksize=4))
    groups = {}
    for name, out, n_left in res:
        g = groups.setdefault(name, dict(frames=0, n_left=0, modes={k: [0, 0, [], 0] for k in out}))
        g["frames"] += 1
        g["n_left"] += n_left
        for k, (same, first, nchg, zsign) in out.items():
            g["modes"][k][3] += 1 if zsign else 0
            if not same:
                g["modes"][k][0] += 1
                g["modes"][k][1] += nchg
                g["modes"][k][2].append(first)
    label = {"0l": "un-fused, lowest index (the build's specification)", "0c": "un-fused, CUDA tree tie rule",
             "1l": "fma(dz,dz,fma(dx,dx,dy*dy)), lowest index", "1c": "fma(dz,dz,fma(dx,dx,dy*dy)), CUDA tree tie rule",
             "2l": "fma(dz,dz,fma(dy,dy,dx*dx)), lowest index", "2c": "fma(dz,dz,fma(dy,dy,dx*dx)), CUDA tree tie rule"}
    names = {"bench": "bench batch: synthetic 64x2048, ids 0..%d" % (a.frames - 1), "example": "reference example.bin (64x2000)",
             "Velodyne64E": "synthetic 64x2000 (Velodyne64E.yaml)", "Velodyne32E": "synthetic 32x2250", "VelodyneVLP16": "synthetic 16x1800"}
    L = ["# FPS: sensitivity of the selected centres to the CUDA binary's unobservable degrees of freedom", "",
         "`python tools_dev/fps_sensitivity.py` (CPU, oracle/orc_fps_modes).  Per frame set and mode: frames in which at least one of the",
         "100 centre COORDINATES differs from the default mode's (un-fused distance, lowest index among equal values), the number of",
         "centres that differ in those frames, and the earliest iteration at which a frame diverges (coordinates compared as values; the",
         "last column counts frames in which a centre of the origin class -- the empty pixels, all at (+-0, +-0, +-0) -- was taken from",
         "another member: same point, other zero signs, no effect on any later stage).  The tie rule can only act on exactly",
         "equal distances between DISTINCT points (equal coordinates give equal centres whichever index wins); a contraction changes",
         "roundings, so two nearly equidistant candidates can swap, after which the two runs select different (equally valid) centre sets.", ""]
    for name in ("bench", "example", "Velodyne64E", "Velodyne32E", "VelodyneVLP16"):
        if name not in groups:
            continue
        g = groups[name]
        L += ["## %s -- %d frame(s), %.0f candidates per frame" % (names[name], g["frames"], g["n_left"] / g["frames"]), "",
              "| mode | frames with >= 1 changed centre | changed centres in those frames | earliest diverging iteration | frames where only the zero signs of an origin centre differ |", "|---|---|---|---|---|"]
        for k in ("0l", "0c", "1l", "1c", "2l", "2c"):
            c, n, firsts, zs = g["modes"][k]
            L.append("| %s | %d of %d | %d | %s | %d |" % (label[k], c, g["frames"], n, min(firsts) if firsts else "-", zs))
        L.append("")
    text = "\n".join(L)
    print(text)
    if a.md:
        open(a.md, "w").write(text)
        json.dump({k: {m: [v[0], v[1], v[3]] for m, v in g["modes"].items()} | {"frames": g["frames"]} for k, g in groups.items()},
                  open(a.md.replace(".md", ".json"), "w"), indent=1)


if __name__ == "__main__":
    main()

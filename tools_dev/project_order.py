#!/usr/bin/env python3
"""Developer experiment (GPU box): the projection (pixel + band kernels, stage entry) on points in random order, in scan order
(one image band per chunk of points: the LDS band counters of the pixel kernel see 64 lanes on one address) and on the
reference's example sweep as stored.   usage: python3 tools_dev/project_order.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import rpcc_amd  # noqa: F401
from rpcc_amd import _lib, ops, synth

dev = torch.device("cuda:0")
B = 256


def timed(xyz, offs, geom, n=8):
    scratch = torch.empty(_lib.lib().rpcc_project_scratch_bytes(xyz.shape[0], B, geom.H * geom.W), dtype=torch.uint8, device=dev)
    ri = None
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ri = ops.project(xyz, offs, geom, ri=ri, scratch=scratch)
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts)), ri


def batch(frames):
    offs = np.zeros(len(frames) + 1, np.int64)
    offs[1:] = np.cumsum([f.shape[0] for f in frames])
    return torch.from_numpy(np.concatenate(frames)).to(dev), torch.from_numpy(offs).to(dev)


hfov, vmax, vmin = 2 * np.pi, 2.0 * np.pi / 180, -24.9 * np.pi / 180
geom = ops.make_geom(64, 2048, hfov, vmax, vmin)
fr = [synth.make_frame(i, 64, 2048).numpy() for i in range(16)]


def scan_order(f):
    row = np.round((f[:, 2] / np.linalg.norm(f, axis=1)) * 400).astype(np.int64)
    return f[np.lexsort((np.arctan2(f[:, 1], f[:, 0]), -row))]


sets = {"synthetic, random order": [fr[i % 16] for i in range(B)],
        "synthetic, scan order": [scan_order(fr[i % 16]) for i in range(B)]}
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "example_64E.npz"))
sets["example sweep as stored (64x2000)"] = [z["xyz"]] * B
ref = {}
for name, frames in sets.items():
    g = ops.make_geom(64, 2000, hfov, vmax, vmin) if "example" in name else geom
    xyz, offs = batch(frames)
    t, ri = timed(xyz, offs, g)
    print("%-40s %8.1f us per %d frames (%.1f M points)" % (name, t, B, xyz.shape[0] / 1e6), flush=True)
    ref[name] = ri
assert torch.equal(ref["synthetic, random order"], ref["synthetic, scan order"])

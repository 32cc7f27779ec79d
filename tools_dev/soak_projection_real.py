#!/usr/bin/env python3
"""Soak of the projection's screened fast path on REAL point distributions (GPU box): the example sweep under random rotations, tilts and scales --
points spread evenly over the pixel, 1.7 % of them inside the margin -- through the record kernels (fast pixel + exact queue) and through the
device-atomic kernels (the exact sequence for every point); the two images must be equal bit for bit.  usage: soak_projection_real.py [variants]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rpcc_amd  # noqa: E402,F401
from rpcc_amd import ops  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda:0")
xyz = torch.from_numpy(np.load(os.path.join(ROOT, "tests", "golden", "example_64E.npz"))["xyz"]).to(dev)
geoms = [(64, 2000, 2.0, -24.9), (64, 2048, 2.0, -24.9), (32, 2250, 10.67, -30.67), (16, 1800, 15.0, -15.0), (80, 2000, 3.0, -25.0)]
rng = np.random.default_rng(606)
t0, bad, pts = time.time(), 0, 0
B = 32
for k0 in range(0, K, B):
    H, W, vmax, vmin = geoms[(k0 // B) % len(geoms)]
    geom = ops.make_geom(H, W, 2 * np.pi, vmax * np.pi / 180, vmin * np.pi / 180)
    frames = []
    for _ in range(B):
        a, t, s = rng.uniform(0, 2 * np.pi), rng.normal(0, 0.02), rng.uniform(0.4, 2.5)
        ca, sa, ct, st = np.cos(a), np.sin(a), np.cos(t), np.sin(t)
        R = torch.tensor([[ca, -sa, 0], [sa, ca, 0], [0, 0, 1]], dtype=torch.float32, device=dev) @ \
            torch.tensor([[ct, 0, st], [0, 1, 0], [-st, 0, ct]], dtype=torch.float32, device=dev)
        frames.append((xyz @ R.T) * np.float32(s))
    pts += B * xyz.shape[0]
    offs = torch.arange(B + 1, dtype=torch.int64, device=dev) * xyz.shape[0]
    cat = torch.cat(frames).contiguous()
    a = ops.project(cat, offs, geom)
    b = ops.project(cat, offs, geom, atomic_path=True)
    bad += int((a.view(torch.int32) != b.view(torch.int32)).sum().item())
print("projection soak on real point distributions: %d variants, %.2e points, %d differing pixels, %.0f s" % (K, pts, bad, time.time() - t0))
sys.exit(1 if bad else 0)

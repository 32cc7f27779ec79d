"""GPU box, library built with -DRPCC_DEVTRACE: cycle stamps of the ground fit on the whole cloud (sweeps without ground returns).
usage: RPCC_EXTRA_FLAGS=-DRPCC_DEVTRACE (build) ; python tools_dev/ransac_phases_wc.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpcc_amd  # noqa: E402,F401
from rpcc_amd import _lib, ops, synth  # noqa: E402

dev = torch.device("cuda:0")
for (H, W, vmax, vmin) in ((64, 2048, 2.0, -24.9), (16, 1800, 15.0, -15.0)):
    geom = ops.make_geom(H, W, 2 * np.pi, vmax * np.pi / 180, vmin * np.pi / 180)
    tm = torch.from_numpy(ops.transform_map(H, W, 2 * np.pi, vmax * np.pi / 180, vmin * np.pi / 180)).to(dev)
    B = 256
    f = synth.make_frame(100, H, W, vmax_deg=vmax, vmin_deg=vmin).numpy()
    f = f[f[:, 2] > -1.45]
    offs = np.arange(B + 1, dtype=np.int64) * f.shape[0]
    xyz = torch.from_numpy(np.tile(f, (B, 1))).to(dev)
    ri = ops.project(xyz, torch.from_numpy(offs).to(dev), geom)
    ops.ground_ransac(ri, tm, 0)
    stamps = torch.zeros(4096 + 16 * 128 * 8, dtype=torch.int64, device=dev)
    _lib.check(_lib.lib().rpcc_debug_stamps(_lib.ptr(stamps)))
    ops.ground_ransac(ri, tm, 0)
    torch.cuda.synchronize()
    _lib.lib().rpcc_debug_stamps(None)
    s = stamps.cpu().numpy()[:64]
    nz = np.flatnonzero(s)
    print("%dx%d whole-cloud ground fit, stamps (cycles since the first):" % (H, W))
    for i in nz:
        print("   slot %2d: %10d" % (i, s[i] - s[nz[0]]))

"""GPU box: share of the points that the projection's screened fast path hands to the exact fdlibm sequence, real sweep against synthetic ones.
usage: python tools_dev/fastpath_fraction.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rpcc_amd  # noqa: E402,F401
from rpcc_amd import ops, synth  # noqa: E402

dev = torch.device("cuda:0")
hf, vmax, vmin = 2 * np.pi, 2.0 * np.pi / 180, -24.9 * np.pi / 180
xyz = np.load(os.path.join(ROOT, "tests", "golden", "example_64E.npz"))["xyz"]
for name, pts, (H, W) in (("real sweep, 64x2000", xyz, (64, 2000)), ("real sweep, 64x2048", xyz, (64, 2048)),
                          ("synthetic 64x2048", synth.make_frame(1, 64, 2048).numpy(), (64, 2048)),
                          ("synthetic 64x2000", synth.make_frame(1, 64, 2000).numpy(), (64, 2000))):
    geom = ops.make_geom(H, W, hf, vmax, vmin)
    sure, bad, slow, dc, dr = ops.project_fastpath_check(torch.from_numpy(np.ascontiguousarray(pts)).to(dev), geom)
    print("%-22s points %7d  certain %7d  to the exact sequence %6d (%.2f %%)  disagreements %d" % (name, pts.shape[0], sure, slow, 100.0 * slow / pts.shape[0], bad))
# where the uncertain ones are: fractional part of the column coordinate
x, y = xyz[:, 0].astype(np.float64), xyz[:, 1].astype(np.float64)
az = np.arctan2(y, x); az[az < 0] += 2 * np.pi
cf = az / (2 * np.pi) * 2000
print("real sweep: histogram of frac(column coordinate), 10 bins:", np.histogram(cf - np.floor(cf), bins=10, range=(0, 1))[0])

#!/usr/bin/env python3
"""Developer tool (library built with -DRPCC_DEVTRACE): how the tiles an FPS iteration has to visit spread over the 8 wavefronts
(each wavefront visits its OWN tiles, two at a time): visit rounds of the slowest wavefront against a perfectly balanced split."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rpcc_amd  # noqa: F401
from rpcc_amd import ops, synth, _lib

dev = torch.device("cuda:0")
B, H, W = 16, 64, 2048
hfov, vmax, vmin = 2 * np.pi, 2.0 * np.pi / 180, -24.9 * np.pi / 180
geom = ops.make_geom(H, W, hfov, vmax, vmin)
tm = torch.from_numpy(ops.transform_map(H, W, hfov, vmax, vmin)).to(dev)
xyz, offs = synth.make_batch(range(B), H, W, device=dev)
ri = ops.project(xyz, offs, geom)
g, inl = ops.ground_ransac(ri, tm, 0)
stamps = torch.zeros(4096 + 16 * 128 * 8, dtype=torch.int64, device=dev)
temp, info, tab = ops.ground_mask(ri, tm, g, 0.1, fps_table=True)
_lib.check(_lib.lib().rpcc_debug_stamps(_lib.ptr(stamps)))
ops.fps_range(ri, tm, temp, info, 100, fps_table=tab)
torch.cuda.synchronize()
_lib.lib().rpcc_debug_stamps(None)
v = stamps.cpu().numpy()[4096:].reshape(16, 128, 8)[:, 2:100]       # [block, iteration, wave]
n = v.sum(2); m = v.max(2)
r_now = np.ceil(m / 2); r_bal = np.ceil(np.ceil(n / 8) / 2)
print("tiles to visit per iteration: mean %.1f, median %d, p90 %d, max %d" % (n.mean(), np.median(n), np.percentile(n, 90), n.max()))
print("most loaded wavefront: mean %.2f tiles; visit rounds (2 tiles each) now: mean %.2f; balanced over 8 wavefronts: mean %.2f" % (m.mean(), r_now.mean(), r_bal.mean()))
for k in range(0, 6):
    print("  rounds == %d: now %5.1f %%   balanced %5.1f %%" % (k, 100 * (r_now == k).mean(), 100 * (r_bal == k).mean()))
print("iterations 2..9 (tiles):", n[:, :8].mean(0).round(1))

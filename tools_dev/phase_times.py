#!/usr/bin/env python3
"""Developer tool: phase stamps (shader clock) of block 0 of the instrumented single-WG-per-frame kernels."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rpcc_amd
from rpcc_amd import ops, synth, _lib

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H, W = 64, 2048
hfov, vmax, vmin = 2*np.pi, 2.0*np.pi/180, -24.9*np.pi/180
geom = ops.make_geom(H, W, hfov, vmax, vmin)
tm = torch.from_numpy(ops.transform_map(H, W, hfov, vmax, vmin)).to(dev)
xyz, offs = synth.make_batch(range(B), H, W, device=dev)
ri = ops.project(xyz, offs, geom)
stamps = torch.zeros(4096 + 16 * 128 * 8, dtype=torch.int64, device=dev)
for which in ("ransac", "fps"):
    stamps.zero_()
    _lib.check(_lib.lib().rpcc_debug_stamps(_lib.ptr(stamps)))
    if which == "ransac":
        g, inl = ops.ground_ransac(ri, tm, 0)
    else:
        temp, info, tab = ops.ground_mask(ri, tm, g, 0.1, fps_table=True)
        ops.fps_range(ri, tm, temp, info, 100, fps_table=tab)
    torch.cuda.synchronize()
    _lib.lib().rpcc_debug_stamps(None)
    s = stamps.cpu().numpy()
    if which == "fps" and s[64 + 8 * 2:].any():
        tr = s[64:64 + 8 * 100].reshape(100, 8)[2:100]
        print("per-iteration trace (block 0, wave 0; cycles after barrier 1): n | test(acc) | issued | arrived | computed | barrier2 | select")
        for j in (0, 1, 2, 10, 30, 50, 70, 90, 97):
            print("   j=%2d  n=%3d  issued %5d  arrived %5d  computed %5d  barrier2 %5d  select %5d" % (j + 2, tr[j, 0], tr[j, 2], tr[j, 3], tr[j, 4], tr[j, 5], tr[j, 6]))
        one = tr[tr[:, 0] <= 16]
        print("   mean over the %d one-round iterations: n %.1f issued %.0f arrived %.0f computed %.0f barrier2 %.0f select %.0f" % ((len(one),) + tuple(one[:, k].mean() for k in (0, 2, 3, 4, 5, 6))))
        two = tr[tr[:, 0] > 16]
        if len(two):
            print("   mean over the %d multi-round iterations: n %.1f issued %.0f arrived %.0f computed %.0f barrier2 %.0f select %.0f" % ((len(two),) + tuple(two[:, k].mean() for k in (0, 2, 3, 4, 5, 6))))
    s = s[:64]
    nz = np.flatnonzero(s)
    print(which, "stamps (cycles since first, ~100MHz or shader clk):")
    base = s[nz[0]] if len(nz) else 0
    for i in nz:
        print("   slot %2d: %10d" % (i, s[i] - base if i < 24 else s[i]))

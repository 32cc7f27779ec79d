#!/usr/bin/env python3
"""Developer tool (library built with -DRPCC_DEVTRACE): cycles per phase of the FPS iteration chain, summed over the 98 iterations,
for every wavefront of block 0.  usage: fps_phases.py [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rpcc_amd  # noqa: F401
from rpcc_amd import ops, synth, _lib

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H, W = 64, 2048
hfov, vmax, vmin = 2 * np.pi, 2.0 * np.pi / 180, -24.9 * np.pi / 180
geom = ops.make_geom(H, W, hfov, vmax, vmin)
tm = torch.from_numpy(ops.transform_map(H, W, hfov, vmax, vmin)).to(dev)
xyz, offs = synth.make_batch(range(B), H, W, device=dev)
ri = ops.project(xyz, offs, geom)
g, inl = ops.ground_ransac(ri, tm, 0)
stamps = torch.zeros(4096 + 16 * 128 * 8, dtype=torch.int64, device=dev)
for rep in range(2):
    temp, info, tab = ops.ground_mask(ri, tm, g, 0.1, fps_table=True)
    stamps.zero_()
    _lib.check(_lib.lib().rpcc_debug_stamps(_lib.ptr(stamps)))
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    ops.fps_range(ri, tm, temp, info, 100, fps_table=tab)
    t1.record()
    torch.cuda.synchronize()
    _lib.lib().rpcc_debug_stamps(None)
s = stamps.cpu().numpy()[64:64 + 64].reshape(8, 8)
print("launch %.1f us; per wavefront of block 0, cycles summed over 98 iterations (readcyclecounter units):" % (t0.elapsed_time(t1) * 1e3))
print("wave |   test |  visit | to-barrier | barrier wait | after barrier+store | tiles visited | iterations with a visit")
for w in range(8):
    print("  %d  | %6d | %6d | %6d | %6d | %6d | %4d | %3d" % (w, s[w, 1], s[w, 2], s[w, 3], s[w, 4], s[w, 5], s[w, 6], s[w, 7]))
vv = stamps.cpu().numpy()[3000:3032].reshape(8, 4)
print("visit rounds per wavefront: cycles issuing loads / waiting for the data / updating, per round:")
for w in range(8):
    r = max(int(vv[w, 3]), 1)
    print("  %d  | rounds %3d | issue %5d | wait %5d | update %5d" % (w, vv[w, 3], vv[w, 0] / r, vv[w, 1] / r, vv[w, 2] / r))
tot = s[:, 1:6].sum(1)
print("sum per wave:", tot, " -> per iteration", (tot / 98).round(0))

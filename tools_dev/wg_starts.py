#!/usr/bin/env python3
"""Developer experiment (GPU box, library built with -DRPCC_DEVTRACE): when do the 256 workgroups of the FPS kernel START and how
long does each RUN -- alone and next to a VALU-bound neighbour kernel looping on another stream?  Separates "the workgroups
wait for a place on a CU" from "the workgroups run slowly".   usage: python3 tools_dev/wg_starts.py"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rpcc_amd  # noqa: F401
from rpcc_amd import ops, synth, _lib
dev = torch.device("cuda:0")
B, H, W, M = 256, 64, 2048, 100
hfov, vmax, vmin = 2 * np.pi, 2.0 * np.pi / 180, -24.9 * np.pi / 180
geom = ops.make_geom(H, W, hfov, vmax, vmin)
tm = torch.from_numpy(ops.transform_map(H, W, hfov, vmax, vmin)).to(dev)
xyz, offs = synth.make_batch(range(B), H, W, device=dev)
ri = ops.project(xyz, offs, geom)
ground, _ = ops.ground_ransac(ri, tm, 0)
temp0, info, tab = ops.ground_mask(ri, tm, ground, 0.1, fps_table=True)
cen_pix, centers = ops.fps_range(ri, tm, temp0.clone(), info, M, fps_table=tab)
ri2, seg2 = ri.clone(), torch.empty((B, H, W), dtype=torch.uint8, device=dev)
stamps = torch.zeros(max(2048 + 2 * B + 64, 4096 + 16 * 128 * 8), dtype=torch.int64, device=dev)
side = torch.cuda.Stream(device=dev)
nb = {"none": None, "assign": lambda: ops.assign(ri2, tm, ground, centers, out=seg2),
      "ground mask": lambda: ops.ground_mask(ri2, tm, ground, 0.1, fps_table=True)}
for name, fn in nb.items():
    stop = [False]
    def spin():
        torch.cuda.set_device(dev)
        with torch.cuda.stream(side):
            while not stop[0]:
                for _ in range(4):
                    fn()
                side.synchronize()
    th = None
    if fn is not None:
        th = threading.Thread(target=spin); th.start(); time.sleep(0.3)
    res = []
    for rep in range(5):
        t = temp0.clone(); stamps.zero_(); torch.cuda.synchronize()
        _lib.check(_lib.lib().rpcc_debug_stamps(_lib.ptr(stamps)))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.fps_range(ri, tm, t, info, M, fps_table=tab, cen_pix=cen_pix, centers=centers); e1.record(); e1.synchronize()
        _lib.lib().rpcc_debug_stamps(None)
        s = stamps.cpu().numpy()[2048:2048 + 2 * B].reshape(B, 2).astype(np.float64) / 100.0   # us
        st, du = s[:, 0] - s[:, 0].min(), s[:, 1] - s[:, 0]
        res.append((e0.elapsed_time(e1) * 1e3, np.percentile(st, [50, 90, 100]), np.percentile(du, [10, 50, 90, 100]), (s[:, 1].max() - s[:, 0].min())))
    if th is not None:
        stop[0] = True; th.join()
    for r in res[1:]:
        print("next to %-12s launch %6.1f us | workgroup START after the first: median %6.1f p90 %6.1f last %6.1f us | RUN time: p10 %6.1f median %6.1f p90 %6.1f max %6.1f us | first start -> last end %6.1f"
              % (name, r[0], r[1][0], r[1][1], r[1][2], r[2][0], r[2][1], r[2][2], r[2][3], r[3]), flush=True)

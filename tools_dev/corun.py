#!/usr/bin/env python3
"""Developer experiment (GPU box): what stretches the one-workgroup-per-frame kernels when batches are in flight?
The FPS kernel (and the ground RANSAC, the band projection) of one 256-frame batch is timed alone and while a second stream
keeps the chip busy with ONE kind of neighbour in a loop: the VALU-bound assign kernel (0.9 TB/s of traffic, 83 % VALU issue),
the HBM-bound pixel kernel (5.3 TB/s, 89 % VALU), the quantiser, the histogram.
usage: python3 tools_dev/corun.py"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import rpcc_amd  # noqa: F401
from rpcc_amd import ops, synth

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
seg = ops.assign(ri, tm, ground, centers)
ws = ops.workspace(B, H * W, M, dev, int(xyz.shape[0]))
model, counts = ops.point_model(ri, seg, ground, M, ws=ws)
torch.cuda.synchronize()

# second copies for the neighbour stream (no buffer shared with the timed kernel is written by the neighbour)
ri2, seg2 = ri.clone(), torch.empty_like(seg)
ws2 = ops.workspace(B, H * W, M, dev, int(xyz.shape[0]))
from rpcc_amd import _lib
scratch2 = torch.empty(_lib.lib().rpcc_project_scratch_bytes(xyz.shape[0], B, H * W), dtype=torch.uint8, device=dev)
ri3 = torch.empty_like(ri)
q2 = torch.zeros((B, H * W), dtype=torch.int16, device=dev)
nnz2 = torch.empty((B,), dtype=torch.int32, device=dev)

neighbours = {
    "none": None,
    "assign (VALU-bound, 0.9 TB/s)": lambda: ops.assign(ri2, tm, ground, centers, out=seg2),
    "project pix+band (HBM-bound)": lambda: ops.project(xyz, offs, geom, ri=ri3, scratch=scratch2),
    "point model hist+scan": lambda: ops.point_model(ri2, seg, ground, M, ws=ws2),
    "predict+quantize (hist+scan+quant)": lambda: ops.predict_quantize(ri2, tm, seg, model, 0.04, M, int16=True, ws=ws2, q_out=q2, nnz_out=nnz2),
    "ground mask": lambda: ops.ground_mask(ri2, tm, ground, 0.1, fps_table=True),
}


def timed_fps(n=6):
    ts = []
    for _ in range(n):
        t = temp0.clone()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.fps_range(ri, tm, t, info, M, fps_table=tab, cen_pix=cen_pix, centers=centers)
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return ts


def timed(fn, n=6):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return ts


targets = {
    "fps": timed_fps,
    "ransac": lambda: timed(lambda: ops.ground_ransac(ri, tm, 0)),
}
side = torch.cuda.Stream(device=dev)
for tname, tfn in targets.items():
    for name, fn in neighbours.items():
        stop = [False]
        count = [0]

        def spin():
            torch.cuda.set_device(dev)
            with torch.cuda.stream(side):
                while not stop[0]:
                    for _ in range(4):
                        fn()
                    count[0] += 4
                    side.synchronize()
        th = None
        if fn is not None:
            th = threading.Thread(target=spin)
            th.start()
            time.sleep(0.3)
        ts = tfn()
        if th is not None:
            stop[0] = True
            th.join()
        torch.cuda.synchronize()
        print("%-7s next to %-38s median %7.1f us  (min %7.1f, max %7.1f)" % (tname, name, float(np.median(ts)), min(ts), max(ts)), flush=True)

#!/usr/bin/env python3
"""Developer experiment (GPU box): when do the batches of a K-step timed region complete?  Three batches in flight from an idle,
synchronised GPU (the bench's timed region): completion time of every batch, the steady period, and what a staggered start
(the second / third stream's first batch delayed by a third / two thirds of a period) changes.
usage: python3 tools_dev/fill_timeline.py [K]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import rpcc_amd  # noqa: F401
from rpcc_amd import ops, synth

dev = torch.device("cuda:0")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, H, W, M, depth = 256, 64, 2048, 100, 3
hfov, vmax, vmin = 2 * np.pi, 2.0 * np.pi / 180, -24.9 * np.pi / 180
geom = ops.make_geom(H, W, hfov, vmax, vmin)
tm = torch.from_numpy(ops.transform_map(H, W, hfov, vmax, vmin)).to(dev)
xyz, offs = synth.make_batch(range(B), H, W, device=dev)
fid = torch.arange(B, dtype=torch.int64, device=dev)
bufs = [ops.BatchBuffers(B, geom, M, dev) for _ in range(depth)]
gms = [torch.zeros((B, 4), dtype=torch.float64, device=dev) for _ in range(depth)]
streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]


timer = ops.FpsTimer() if os.environ.get("FILL_TIMER") else None   # (FILL_TIMER=1: with the bench's HIP events around every FPS launch)
if timer is not None:
    timer.reserve(4096)


def run(k):
    ops.compress_batch(xyz, offs, tm, gms[k], bufs[k], ground_threshold=0.1, acc=0.02, ground_seed=0, frame_ids=fid, timer=timer)


for k in range(depth):
    with torch.cuda.stream(streams[k]):
        run(k)
torch.cuda.synchronize()
clock_hz = 100e6   # torch.cuda._sleep counts s_memtime / wall-clock-ish cycles; calibrated below
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record(); torch.cuda._sleep(10_000_000); t1.record(); torch.cuda.synchronize()
cyc_per_ms = 10_000_000 / t0.elapsed_time(t1)


def region(stagger_ms, steps):
    for _ in range(6):
        for k in range(depth):
            with torch.cuda.stream(streams[k]):
                run(k)
    torch.cuda.synchronize()
    if timer is not None:
        timer.read()
    start = torch.cuda.Event(enable_timing=True)
    ev = []
    w0 = time.perf_counter()
    start.record()
    for s in streams:
        s.wait_event(start)
    for n in range(steps):
        k = n % depth
        with torch.cuda.stream(streams[k]):
            if n < depth and stagger_ms and k:
                torch.cuda._sleep(int(stagger_ms * k * cyc_per_ms))
            run(k)
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            ev.append(e)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - w0) * 1e3
    done = [start.elapsed_time(e) for e in ev]
    return wall, done


for stagger in ((0.0,) if os.environ.get("FILL_ONLY0") else (0.0, 0.15, 0.27, 0.40)):
    walls = []
    for rep in range(3):
        wall, done = region(stagger, K)
        walls.append(wall)
    d = np.array(sorted(done))
    per = np.diff(d)
    print("stagger %.2f ms: wall %.3f / %.3f / %.3f ms for %d steps = %.4f ms per step; completions (ms): %s" %
          (stagger, walls[0], walls[1], walls[2], K, min(walls) / K, " ".join("%.2f" % v for v in d[:8])), "... last", "%.2f" % d[-1], flush=True)
    print("      gaps between completions: first six %s, median of the rest %.3f" % (" ".join("%.2f" % v for v in per[:6]), float(np.median(per[6:])) if len(per) > 6 else 0.0))

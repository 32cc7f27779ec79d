#!/usr/bin/env python3
"""Developer tool (GPU box): FPS kernel alone, fused step serial and with three batches in flight, for the library as built.
usage: [RPCC_EXTRA_FLAGS="-DFPS_GROUP=1 ..."] python3 tools_dev/fps_time.py [tag]   (rebuilds the library first when flags are given)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rpcc_amd
from rpcc_amd import build as b
if os.environ.get("RPCC_EXTRA_FLAGS") is not None:
    b.build(force=True)
from rpcc_amd import ops, synth

tag = sys.argv[1] if len(sys.argv) > 1 else os.environ.get("RPCC_EXTRA_FLAGS", "default")
dev = torch.device("cuda:0")
B, H, W, M = 256, 64, 2048, 100
hfov, vmax, vmin = 2 * np.pi, 2.0 * np.pi / 180, -24.9 * np.pi / 180
geom = ops.make_geom(H, W, hfov, vmax, vmin)
tm = torch.from_numpy(ops.transform_map(H, W, hfov, vmax, vmin)).to(dev)
xyz, offs = synth.make_batch(range(B), H, W, device=dev)
ri = ops.project(xyz, offs, geom)
g, _ = ops.ground_ransac(ri, tm, 0)

def ev_time(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

temps = []
def fps_only():
    temp, info, tab = ops.ground_mask(ri, tm, g, 0.1, fps_table=True)
    ops.fps_range(ri, tm, temp, info, M, fps_table=tab)
def mask_only():
    ops.ground_mask(ri, tm, g, 0.1, fps_table=True)
t_both = ev_time(fps_only, 10); t_mask = ev_time(mask_only, 10)
depth = 3
bufs = [ops.BatchBuffers(B, geom, M, dev) for _ in range(depth)]
gms = [torch.zeros((B, 4), dtype=torch.float64, device=dev) for _ in range(depth)]
streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
def serial():
    ops.compress_batch(xyz, offs, tm, gms[0], bufs[0], ground_seed=0)
t_serial = ev_time(serial, 10)
for k in range(depth):
    with torch.cuda.stream(streams[k]):
        ops.compress_batch(xyz, offs, tm, gms[k], bufs[k], ground_seed=0)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 60
for s in range(N):
    with torch.cuda.stream(streams[s % depth]):
        ops.compress_batch(xyz, offs, tm, gms[s % depth], bufs[s % depth], ground_seed=0)
torch.cuda.synchronize()
t_pipe = (time.perf_counter() - t0) / N * 1e6
# configs[2]: non-uniform + plane, three in flight
gb = [ops.BatchBuffers(B, geom, M, dev, general=True) for _ in range(depth)]
nu = ops.nonuniform_cfg(0.04)
def c2(k):
    ops.compress_batch(xyz, offs, tm, gms[k], gb[k], ground_seed=0, model_method="plane", nonuniform=nu)
for k in range(depth):
    with torch.cuda.stream(streams[k]):
        c2(k)
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in range(30):
    with torch.cuda.stream(streams[s % depth]):
        c2(s % depth)
torch.cuda.synchronize()
t_c2 = (time.perf_counter() - t0) / 30 * 1e6
print("VARIANT %-40s fps+mask %7.1f us  (mask %6.1f, fps ~%6.1f)  step serial %7.1f us  pipelined %7.1f us  config2 pipelined %7.1f us" % (tag, t_both, t_mask, t_both - t_mask, t_serial, t_pipe, t_c2), flush=True)

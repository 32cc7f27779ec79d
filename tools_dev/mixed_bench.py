#!/usr/bin/env python3
"""Developer benchmark for configs[4]: a mixed 64E / 32E / VLP16 batch (non-uniform + plane), device part only: the three
geometry groups one after the other on one stream vs overlapped on three streams (what MixedBatchCompressor does)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rpcc_amd  # noqa: F401
from rpcc_amd import synth, dataset
from rpcc_amd.pipeline import BatchCompressor
from oracle import oracle as orc  # geometry table only

per = int(sys.argv[1]) if len(sys.argv) > 1 else 85
dev = torch.device("cuda:0")
names = ["Velodyne64E", "Velodyne32E", "VelodyneVLP16"]
groups = {}
for n in names:
    gd = orc.GEOMS[n]
    T = dataset.build_dataset(lidar_type=n).PCTransformer
    xyz, offs = synth.make_batch(range(per), gd["H"], gd["W"], device=dev, vmax_deg=gd["vmax_deg"], vmin_deg=gd["vmin_deg"])
    groups[n] = (BatchCompressor(T, uniform=False, model_method="plane", seed=1), xyz, offs, torch.cuda.Stream(device=dev))
for bc, xyz, offs, _ in groups.values():
    bc.compress_device(xyz, offs)
torch.cuda.synchronize()
R = 5
t0 = time.perf_counter()
for _ in range(R):
    for bc, xyz, offs, _ in groups.values():
        bc.compress_device(xyz, offs)
torch.cuda.synchronize()
t_seq = (time.perf_counter() - t0) / R
for bc, xyz, offs, st in groups.values():   # warm the per-stream allocator pools
    with torch.cuda.stream(st):
        bc.compress_device(xyz, offs)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(R):
    for bc, xyz, offs, st in groups.values():
        with torch.cuda.stream(st):
            bc.compress_device(xyz, offs)
torch.cuda.synchronize()
t_ovl = (time.perf_counter() - t0) / R
n = 3 * per
print("mixed batch of %d frames (non-uniform + plane): groups in sequence %.3f ms (%.0f frames/s), overlapped on 3 streams %.3f ms (%.0f frames/s)"
      % (n, t_seq * 1e3, n / t_seq, t_ovl * 1e3, n / t_ovl))

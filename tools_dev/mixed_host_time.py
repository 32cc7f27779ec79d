"""GPU box: is the mixed-lidar secondary (bench.run_mixed) bound by the host's issue rate?  Times the issue loop (no wait) and the whole
region for 85 + 85 + 85 sweeps per mixed batch, and splits one group's call into its parts.  usage: python tools_dev/mixed_host_time.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpcc_amd  # noqa: E402,F401
from oracle import oracle as orc  # noqa: E402
from rpcc_amd import dataset, ops, synth  # noqa: E402
from rpcc_amd.pipeline import BatchCompressor  # noqa: E402

dev = torch.device("cuda:0")
per, reps = int(os.environ.get("PER", 85)), 48
for slots in (1, 3):
    groups = []
    for n in ("Velodyne64E", "Velodyne32E", "VelodyneVLP16"):
        gd = orc.GEOMS[n]
        T = dataset.build_dataset(lidar_type=n, device=str(dev)).PCTransformer
        ids = list(range(3000, 3000 + per))
        xyz, offs = synth.make_batch(ids, gd["H"], gd["W"], device=dev, vmax_deg=gd["vmax_deg"], vmin_deg=gd["vmin_deg"])
        sl = [(BatchCompressor(T, accuracy=0.02, uniform=False, model_method="plane", seed=1), torch.cuda.Stream(device=dev)) for _ in range(slots)]
        groups.append((n, sl, xyz, offs, torch.as_tensor(np.asarray(ids, np.int64), device=dev)))

    def mixed_batch(r):
        for n, sl, xyz, offs, fid in groups:
            bc, st = sl[r % slots]
            with torch.cuda.stream(st):
                bc.compress_device(xyz, offs, frame_ids=fid)
    for r in range(2 * slots):
        mixed_batch(r)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(reps):
        mixed_batch(r)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("slots %d: issue %.3f ms per mixed batch, whole region %.3f ms per mixed batch (%.0f frames/s)" %
          (slots, (t1 - t0) / reps * 1e3, (t2 - t0) / reps * 1e3, 3 * per * reps / (t2 - t0)), flush=True)

# the parts of one group's call, host time only (the device is idle: nothing waits)
n, sl, xyz, offs, fid = groups[0]
bc, st = sl[0]
B = offs.numel() - 1
buf = bc._buffers(B)
torch.cuda.synchronize()


def timeit(fn, k=200):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(k):
        fn()
    dt = (time.perf_counter() - t) / k * 1e6
    torch.cuda.synchronize()
    return dt


g = torch.zeros((B, 4), dtype=torch.float64, device=dev)
nu = ops.nonuniform_cfg(bc.acc, bc.cfg)
print("torch.zeros((B,4)) %.1f us" % timeit(lambda: torch.zeros((B, 4), dtype=torch.float64, device=dev)))
print("nonuniform_cfg %.1f us" % timeit(lambda: ops.nonuniform_cfg(bc.acc, bc.cfg)))
print("compress_batch (general) %.1f us" % timeit(lambda: ops.compress_batch(xyz, offs, bc.T.tm_dev, g, buf, bc.ground_threshold, bc.acc, ground_seed=1, frame_ids=fid,
                                                                              model_method="plane", angle_threshold=75, plane_seed=1, nonuniform=nu), 50))
print("contour_encode %.1f us" % timeit(lambda: ops.contour_encode(buf.seg, bc.M, ws=bc._codec_ws), 50))
print("compress_device %.1f us" % timeit(lambda: bc.compress_device(xyz, offs, frame_ids=fid), 50))
with torch.cuda.stream(st):
    print("compress_device under a stream context %.1f us" % timeit(lambda: bc.compress_device(xyz, offs, frame_ids=fid), 50))

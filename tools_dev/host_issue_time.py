"""GPU box: how long does the HOST take to issue one call of the batched path (Python wrapper + ctypes + the library's launches)?
A mixed batch of 85 + 85 + 85 sweeps is three such calls (+ three contour-codec calls) per 1.2 ms: host-bound or not?
usage: python tools_dev/host_issue_time.py"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpcc_amd  # noqa: E402,F401
from rpcc_amd import dataset, synth  # noqa: E402
from rpcc_amd.pipeline import BatchCompressor  # noqa: E402
from oracle import oracle as orc  # noqa: E402

dev = torch.device("cuda:0")
gd = orc.GEOMS["VelodyneVLP16"]
T = dataset.build_dataset(lidar_type="VelodyneVLP16", device=str(dev)).PCTransformer
ids = list(range(85))
xyz, offs = synth.make_batch(ids, gd["H"], gd["W"], device=dev, vmax_deg=gd["vmax_deg"], vmin_deg=gd["vmin_deg"])
fid = torch.as_tensor(np.asarray(ids, np.int64), device=dev)
for kw, name in ((dict(uniform=False, model_method="plane"), "non-uniform + plane"), (dict(), "uniform + point")):
    bc = BatchCompressor(T, accuracy=0.02, seed=1, **kw)
    for _ in range(3):
        bc.compress_device(xyz, offs, frame_ids=fid)
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        bc.compress_device(xyz, offs, frame_ids=fid)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("%s: host issue %.1f us per call, device-complete %.1f us per call" % (name, t_issue / n * 1e6, t_all / n * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    bc.compress_device(xyz, offs, frame_ids=fid)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)

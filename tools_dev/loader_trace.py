#!/usr/bin/env python3
"""Developer tool: device-side timeline of loader.StreamingCompressor (HIP events around H2D, compute, D2H per batch)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rpcc_amd  # noqa: F401
from rpcc_amd import synth, loader
from rpcc_amd.transformer import PCTransformer
from rpcc_amd.pipeline import BatchCompressor
B = 256
T = PCTransformer(dict(HORIZONTAL_FOV=360, VERTICAL_ANGLE_MAX=2.0, VERTICAL_ANGLE_MIN=-24.9, RANGE_IMAGE_HEIGHT=64, RANGE_IMAGE_WIDTH=2048))
base = [synth.make_frame(i, 64, 2048, device="cuda:0").cpu().numpy() for i in range(B)]
bc = BatchCompressor(T, seed=1)
sc = loader.StreamingCompressor(bc, batch=B, depth=int(sys.argv[1]) if len(sys.argv) > 1 else 4, workers=16)
ev = []
orig = sc._enqueue
def traced(slot, npts):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    th0 = time.perf_counter()
    with torch.cuda.stream(sc.copy_stream):
        e[0].record(sc.copy_stream)
    orig(slot, npts)
    with torch.cuda.stream(sc.copy_stream):
        e[1].record(sc.copy_stream)
    with torch.cuda.stream(sc.compute_stream):
        e[3].record(sc.compute_stream)
    ev.append((th0, e))
sc._enqueue = traced
def batches(n):
    for k in range(n):
        yield base, None
sc.run(batches(3), entropy=False)
ev.clear()
t0 = time.perf_counter()
sc.run(batches(10), entropy=False)
torch.cuda.synchronize()
e0 = ev[0][1][0]
for k, (th, e) in enumerate(ev):
    print("batch %2d: host enqueue at %6.2f ms | device: H2D start %6.2f end %6.2f | all done %6.2f" %
          (k, (th - t0) * 1e3, e0.elapsed_time(e[0]), e0.elapsed_time(e[1]), e0.elapsed_time(e[3])))

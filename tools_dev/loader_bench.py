#!/usr/bin/env python3
"""Developer benchmark (GPU box): loader.StreamingCompressor end to end from frames in host memory -- staging into pinned
memory, H2D, device part, D2H of the packed payload -- without and with the entropy coder (bzip2 on the pool's threads).
usage: python3 tools_dev/loader_bench.py [batches] [workers]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rpcc_amd  # noqa: F401
from rpcc_amd import synth
from rpcc_amd.transformer import PCTransformer
from rpcc_amd.pipeline import BatchCompressor
from rpcc_amd.loader import StreamingCompressor
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 12
workers = int(sys.argv[2]) if len(sys.argv) > 2 else min(64, os.cpu_count() or 8)
B = 256
T = PCTransformer(dict(HORIZONTAL_FOV=360, VERTICAL_ANGLE_MAX=2.0, VERTICAL_ANGLE_MIN=-24.9, RANGE_IMAGE_HEIGHT=64, RANGE_IMAGE_WIDTH=2048))
base = [synth.make_frame(i, 64, 2048, device="cuda:0").cpu().numpy() for i in range(B)]
bc = BatchCompressor(T, seed=1)
for w in sorted({8, 16, workers}):
    sc = StreamingCompressor(bc, batch=B, depth=4, workers=w)
    def batches(n):
        for k in range(n):
            yield base, list(range(k * B, k * B + B))
    sc.run(batches(2), entropy=False)
    for k in sc.prof: sc.prof[k] = 0.0
    t0 = time.perf_counter(); n = sc.run(batches(NB), entropy=False); torch.cuda.synchronize(); t1 = time.perf_counter() - t0
    print("workers %3d: %d frames from host memory, no entropy coder: %.3f s = %.0f frames/s" % (w, n, t1, n / t1), flush=True)
    print("             host ms per batch: " + ", ".join("%s %.2f" % (k, v / NB * 1e3) for k, v in sc.prof.items()), flush=True)
    # staging alone (host copies into pinned memory) and H2D alone
    t0 = time.perf_counter()
    for k in range(4):
        npts = sc._stage(sc.slots[0], base, None)
    ts = (time.perf_counter() - t0) / 4
    t0 = time.perf_counter()
    for k in range(4):
        sc.slots[0].xyz_dev[:npts].copy_(sc.slots[0].xyz_pin[:npts], non_blocking=True)
    torch.cuda.synchronize(); th = (time.perf_counter() - t0) / 4
    print("             staging %.1f ms per batch (%.1f GB/s), H2D %.1f ms per batch (%.1f GB/s)" % (ts * 1e3, npts * 12 / ts / 1e9, th * 1e3, npts * 12 / th / 1e9), flush=True)
sc = StreamingCompressor(bc, batch=B, depth=4, workers=workers)
t0 = time.perf_counter(); n = sc.run(batches(3), entropy=True); t1 = time.perf_counter() - t0
print("workers %3d: %d frames with bzip2 + container: %.3f s = %.0f frames/s" % (workers, n, t1, n / t1))

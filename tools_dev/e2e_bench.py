#!/usr/bin/env python3
"""Developer benchmark: BatchCompressor.compress end to end (H2D of the points, device part, D2H of the packed streams, bzip2
per frame, container) for 256 synthetic 64x2048 frames -- entropy coding serial vs on a thread pool."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ThreadPoolExecutor
import numpy as np, torch
import rpcc_amd  # noqa: F401
from rpcc_amd import synth
from rpcc_amd.transformer import PCTransformer
from rpcc_amd.pipeline import BatchCompressor
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = PCTransformer(dict(HORIZONTAL_FOV=360, VERTICAL_ANGLE_MAX=2.0, VERTICAL_ANGLE_MIN=-24.9, RANGE_IMAGE_HEIGHT=64, RANGE_IMAGE_WIDTH=2048))
frames = [synth.make_frame(i, 64, 2048).numpy() for i in range(B)]
bc = BatchCompressor(T, seed=1)
bc.compress(frames[:8])
t0 = time.perf_counter(); blobs = bc.compress(frames); t1 = time.perf_counter() - t0
with ThreadPoolExecutor(os.cpu_count() or 8) as pool:
    bc.compress(frames[:8], pool=pool)
    t0 = time.perf_counter(); blobs2 = bc.compress(frames, pool=pool); t2 = time.perf_counter() - t0
assert blobs == blobs2
print("%d frames end to end: entropy coder serial %.2f s (%.0f frames/s), on %d threads %.2f s (%.0f frames/s); %.1f KB per frame"
      % (B, t1, B / t1, os.cpu_count(), t2, B / t2, sum(map(len, blobs)) / B / 1024))
with ThreadPoolExecutor(os.cpu_count() or 8) as pool:
    t0 = time.perf_counter(); ctx = bc.submit(frames); torch.cuda.synchronize(); ts = time.perf_counter() - t0
    t0 = time.perf_counter(); bc.collect(ctx, pool=pool); tc = time.perf_counter() - t0
print("split: submit (concatenate + H2D + device part) %.3f s, collect (D2H + entropy coding + container) %.3f s" % (ts, tc))

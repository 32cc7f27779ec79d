#!/usr/bin/env python3
"""Developer benchmark (GPU box): StreamingCompressor end to end WITH the entropy coder (bzip2 + container) against the
number of host threads.  usage: loader_entropy.py [batches] [workers ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rpcc_amd  # noqa: F401
from rpcc_amd import synth
from rpcc_amd.transformer import PCTransformer
from rpcc_amd.pipeline import BatchCompressor
from rpcc_amd.loader import StreamingCompressor
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ws = [int(x) for x in sys.argv[2:]] or [32, 64, 128]
B = 256
T = PCTransformer(dict(HORIZONTAL_FOV=360, VERTICAL_ANGLE_MAX=2.0, VERTICAL_ANGLE_MIN=-24.9, RANGE_IMAGE_HEIGHT=64, RANGE_IMAGE_WIDTH=2048))
xyz, offs = synth.make_batch(range(B), 64, 2048, device="cuda:0")
x = xyz.cpu().numpy(); o = offs.cpu().numpy()
base = [x[o[i]:o[i + 1]] for i in range(B)]
bc = BatchCompressor(T, seed=1)
def batches(n):
    for k in range(n):
        yield base, list(range(k * B, k * B + B))
import threading
enc_t = [0.0, 0]
lock = threading.Lock()
_orig = StreamingCompressor._encode_chunk
def timed(self, payload, lo, hi):
    t = time.perf_counter(); r = _orig(self, payload, lo, hi); dt = time.perf_counter() - t
    with lock:
        enc_t[0] += dt; enc_t[1] += hi - lo
    return r
StreamingCompressor._encode_chunk = timed
for w in ws:
    enc_t[0] = 0.0; enc_t[1] = 0
    sc = StreamingCompressor(bc, batch=B, depth=4, workers=w)
    sc.run(batches(2), entropy=True)
    nbytes = [0]
    t0 = time.perf_counter(); n = sc.run(batches(NB), sink=lambda k, res: nbytes.__setitem__(0, nbytes[0] + sum(len(r) for r in res)), entropy=True); t1 = time.perf_counter() - t0
    print("workers %3d: %d frames with bzip2 + container: %.2f s = %.0f frames/s, %.1f KB per frame; inside the encode tasks %.1f ms per frame (thread time)"
          % (w, n, t1, n / t1, nbytes[0] / n / 1e3, enc_t[0] / max(enc_t[1], 1) * 1e3), flush=True)

"""GPU box: what each stage of the batch costs IN FLIGHT -- bench.py's three free-running streams with one group of stages left out of every call
(the later stages run on the valid results an earlier full call left in the buffers, so every kernel does its usual work).  The difference to the
full call is the stage's marginal share of the step, the most a cheaper or merged form of it could gain.
usage: python tools_dev/stage_marginals.py [steps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpcc_amd  # noqa: E402,F401
from rpcc_amd import ops, synth  # noqa: E402

dev = torch.device("cuda:0")
H, W, M, B = 64, 2048, 100, 256
hfov, vmax, vmin = 2 * np.pi, 2.0 * np.pi / 180, -24.9 * np.pi / 180
geom = ops.make_geom(H, W, hfov, vmax, vmin)
tm = torch.from_numpy(ops.transform_map(H, W, hfov, vmax, vmin)).to(dev)
ids = list(range(B))
xyz, offs = synth.make_batch(ids, H, W, device=dev)
fid = torch.as_tensor(np.asarray(ids, np.int64), device=dev)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
D = 3
bufs = [ops.BatchBuffers(B, geom, M, dev) for _ in range(D)]
gms = [torch.zeros((B, 4), dtype=torch.float64, device=dev) for _ in range(D)]
kw = dict(ground_seed=0, frame_ids=fid)
S = ops
ALL = S.STAGE_PROJECT | S.STAGE_GROUND | S.STAGE_MASK | S.STAGE_FPS | S.STAGE_LABELS | S.STAGE_PLANES | S.STAGE_QUANTISE
TAIL = S.STAGE_LABELS | S.STAGE_PLANES | S.STAGE_QUANTISE
# (the ground fit is never run without the projection before it: its candidate counts come from the band kernel of the same call)
cases = [("full call", ALL),
         ("without the ground fit (planes of the previous call)", ALL & ~S.STAGE_GROUND),
         ("without mask + FPS (temp / centres of the previous call)", ALL & ~(S.STAGE_MASK | S.STAGE_FPS)),
         ("without the FPS only (assign's reach from the first pass: more survivors)", ALL & ~S.STAGE_FPS),
         ("without assign .. quantiser", ALL & ~TAIL),
         ("without projection + ground fit", ALL & ~(S.STAGE_PROJECT | S.STAGE_GROUND)),
         ("projection + ground fit only", S.STAGE_PROJECT | S.STAGE_GROUND),
         ("mask + FPS only", S.STAGE_MASK | S.STAGE_FPS),
         ("assign .. quantiser only", TAIL)]


def run(mask, n):
    st = [torch.cuda.Stream(device=dev) for _ in range(D)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(n):
        with torch.cuda.stream(st[t % D]):
            ops.compress_batch_stages(mask, xyz, offs, tm, gms[t % D], bufs[t % D], **kw)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for k in range(D):
    ops.compress_batch(xyz, offs, tm, gms[k], bufs[k], **kw)
torch.cuda.synchronize()
for rep in range(3):
    for name, mask in cases:
        for k in range(D):   # valid state for the stages left out
            ops.compress_batch(xyz, offs, tm, gms[k], bufs[k], **kw)
        run(mask, 30)
        dt = run(mask, steps)
        print("%-76s %.4f ms per step" % (name, dt * 1e3), flush=True)

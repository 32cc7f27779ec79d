"""GPU box, library built with -DRPCC_DEVTRACE: cycle stamps of the ground fit inside the fused call (candidate counts and bytes handed over by the band kernel).
usage: (build with RPCC_EXTRA_FLAGS=-DRPCC_DEVTRACE) python tools_dev/ransac_phases_fused.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpcc_amd  # noqa: E402,F401
from rpcc_amd import _lib, ops, synth  # noqa: E402

dev = torch.device("cuda:0")
H, W, M, B = 64, 2048, 100, 256
hf, vmax, vmin = 2 * np.pi, 2.0 * np.pi / 180, -24.9 * np.pi / 180
geom = ops.make_geom(H, W, hf, vmax, vmin)
tm = torch.from_numpy(ops.transform_map(H, W, hf, vmax, vmin)).to(dev)
ids = list(range(B))
xyz, offs = synth.make_batch(ids, H, W, device=dev)
buf = ops.BatchBuffers(B, geom, M, dev)
gms = torch.zeros((B, 4), dtype=torch.float64, device=dev)
fid = torch.as_tensor(np.asarray(ids, np.int64), device=dev)
kw = dict(ground_seed=0, frame_ids=fid)
ops.compress_batch(xyz, offs, tm, gms, buf, **kw)
torch.cuda.synchronize()
stamps = torch.zeros(4096 + 16 * 128 * 8, dtype=torch.int64, device=dev)
ops.compress_batch_stages(ops.STAGE_PROJECT, xyz, offs, tm, gms, buf, **kw)
torch.cuda.synchronize()
_lib.check(_lib.lib().rpcc_debug_stamps(_lib.ptr(stamps)))
ops.compress_batch_stages(ops.STAGE_GROUND, xyz, offs, tm, gms, buf, **kw)
torch.cuda.synchronize()
_lib.lib().rpcc_debug_stamps(None)
s = stamps.cpu().numpy()[:64]
nz = np.flatnonzero(s)
names = {0: "start", 8: "kept pixels noted", 1: "counts known", 2: "list complete", 7: "planes fitted", 3: "scored", 4: "centroid", 5: "summed", 6: "end"}
print("ground fit inside the fused call (block 0), cycles since the first stamp:")
for i in sorted(nz, key=lambda k: s[k]):
    print("   %-14s %10d" % (names.get(int(i), "slot %d" % i), s[i] - s[nz].min()))

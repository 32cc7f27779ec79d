import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import rpcc_amd
from rpcc_amd import ops, synth
dev = torch.device("cuda:0")
H, W, M, B = 64, 2048, 100, 256
hfov, vmax, vmin = 2 * np.pi, 2.0 * np.pi / 180, -24.9 * np.pi / 180
geom = ops.make_geom(H, W, hfov, vmax, vmin)
tm = torch.from_numpy(ops.transform_map(H, W, hfov, vmax, vmin)).to(dev)
ids = list(range(B))
xyz, offs = synth.make_batch(ids, H, W, device=dev)
fid = torch.as_tensor(np.asarray(ids, np.int64), device=dev)
D = 4
bufs = [ops.BatchBuffers(B, geom, M, dev) for _ in range(D)]
gms = [torch.zeros((B, 4), dtype=torch.float64, device=dev) for _ in range(D)]
kw = dict(ground_seed=0, frame_ids=fid)
S = ops
def run(mask, n, depth, pre=0):
    st = [torch.cuda.Stream(device=dev) for _ in range(depth)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(n):
        with torch.cuda.stream(st[t % depth]):
            if pre: ops.compress_batch_stages(pre, xyz, offs, tm, gms[t % depth], bufs[t % depth], **kw)
            ops.compress_batch_stages(mask, xyz, offs, tm, gms[t % depth], bufs[t % depth], **kw)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n
for k in range(D): ops.compress_batch(xyz, offs, tm, gms[k], bufs[k], **kw)
torch.cuda.synchronize()
# the FPS needs the mask before it (temp is its state): mask + FPS per step; mask alone for the difference
for name, mask in (("mask", S.STAGE_MASK), ("mask + FPS", S.STAGE_MASK | S.STAGE_FPS), ("projection", S.STAGE_PROJECT), ("projection + ground fit", S.STAGE_PROJECT | S.STAGE_GROUND),
                   ("assign+hist+scan", S.STAGE_LABELS), ("quantiser", S.STAGE_QUANTISE)):
    for depth in (1, 2, 3, 4):
        run(mask, 20, depth)
        print("%-26s depth %d: %.4f ms per step" % (name, depth, run(mask, 150, depth) * 1e3), flush=True)

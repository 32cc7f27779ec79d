import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import rpcc_amd
from rpcc_amd import ops, synth
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
B=256; dev=torch.device("cuda:0"); H,W,M=64,2048,100
hfov, vmax, vmin = 2*np.pi, 2.0*np.pi/180, -24.9*np.pi/180
geom = ops.make_geom(H, W, hfov, vmax, vmin)
tm = torch.from_numpy(ops.transform_map(H, W, hfov, vmax, vmin)).to(dev)
xyz, offs = synth.make_batch(range(B), H, W, device=dev)
buf = ops.BatchBuffers(B, geom, M, dev, max_points=xyz.shape[0])
g = torch.zeros((B, 4), dtype=torch.float64, device=dev)
ops.compress_batch(xyz, offs, tm, g, buf, ground_seed=0); torch.cuda.synchronize()
for ang in (89.9, 75, 40, 0.001):
    m = ops.plane_model(buf.ri, tm, buf.seg, M, angle_threshold=ang, ground=g)
    frac = float((m[:, 2:, :3] != 0).any(-1).float().mean())
    print("angle %5.1f  plane rows %.2f  %.3f ms" % (ang, frac, timeit(lambda: ops.plane_model(buf.ri, tm, buf.seg, M, angle_threshold=ang, ground=g))))
c = ops.plane_model(buf.ri, tm, buf.seg, M, angle_threshold=75, ground=g, want_counts=True)[1][:, 2:].flatten().float()
print("label sizes: mean %.0f  <30: %.2f  quantiles 50/90/99/99.9/max: %s" % (c.mean(), (c < 30).float().mean(), [int(torch.quantile(c, q)) for q in (0.5, 0.9, 0.99, 0.999, 1.0)]))

import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import rpcc_amd
from rpcc_amd import ops, synth
def timeit(fn, n=10):
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
print("features %.3f ms" % timeit(lambda: ops.extract_features(buf.ri, buf.seg)))
feat, kp = ops.extract_features(buf.ri, buf.seg)
lacc = (np.array([0.04] * 4) + np.array([0, 0.02, 0.04, 0.06])).astype(np.float32)
print("salience %.3f ms" % timeit(lambda: ops.salience(buf.seg, kp, [30, 10, 3, 0], lacc, 2, M)))

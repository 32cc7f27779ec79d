#!/usr/bin/env python3
"""Developer micro-benchmark of the rows outside the fused uniform path (a9, a12, a13, f1, f3) at B frames."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
H, W, M = 64, 2048, 100
hfov, vmax, vmin = 2*np.pi, 2.0*np.pi/180, -24.9*np.pi/180
geom = ops.make_geom(H, W, hfov, vmax, vmin)
tm = torch.from_numpy(ops.transform_map(H, W, hfov, vmax, vmin)).to(dev)
xyz, offs = synth.make_batch(range(B), H, W, device=dev)
buf = ops.BatchBuffers(B, geom, M, dev, max_points=xyz.shape[0])
g = torch.zeros((B, 4), dtype=torch.float64, device=dev)
ops.compress_batch(xyz, offs, tm, g, buf, ground_seed=0)
torch.cuda.synchronize()
ri, seg, model = buf.ri, buf.seg, buf.model
print("B=%d" % B)
print("plane_model       %.3f ms" % timeit(lambda: ops.plane_model(ri, tm, seg, M, ground=g)))
print("extract_features  %.3f ms" % timeit(lambda: ops.extract_features(ri, seg)))
feat, kp = ops.extract_features(ri, seg)
lacc = (np.array([0.04] * 4) + np.array([0, 0.02, 0.04, 0.06])).astype(np.float32)
print("salience          %.3f ms" % timeit(lambda: ops.salience(seg, kp, [30, 10, 3, 0], lacc, 2, M)))
sal, label_acc = ops.salience(seg, kp, [30, 10, 3, 0], lacc, 2, M)
print("predict_q (nonuni) %.3f ms" % timeit(lambda: ops.predict_quantize(ri, tm, seg, model, 0.04, M, int16=True, ws=buf.ws, label_acc=label_acc)))
cws = ops.codec_workspace(B, H * W, M, dev)
print("contour_encode    %.3f ms" % timeit(lambda: ops.contour_encode(seg, M, ws=cws)))
bits, seq, nseq = ops.contour_encode(seg, M, ws=cws)
print("contour_decode    %.3f ms" % timeit(lambda: ops.contour_decode(bits, seq, H, W, M, ws=cws)))
q, nnz, _ = ops.predict_quantize(ri, tm, seg, model, 0.04, M, int16=True, ws=buf.ws)
print("decode            %.3f ms" % timeit(lambda: ops.decode(seg, q, model, tm, 0.04, want_points=True, ws=cws)))
print("backproject       %.3f ms" % timeit(lambda: ops.backproject(ri, tm)))
print("intra_predict     %.3f ms" % timeit(lambda: ops.intra_predict(seg, model, tm)))

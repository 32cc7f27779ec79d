#!/usr/bin/env python3
"""Developer micro-benchmark: per-stage timing of the HIP path on a synthetic batch (torch events)."""
import sys, os, time
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

def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dev = torch.device("cuda:0")
    H, W = 64, 2048
    hfov, vmax, vmin = 2*np.pi, 2.0*np.pi/180, -24.9*np.pi/180
    geom = ops.make_geom(H, W, hfov, vmax, vmin)
    tm = torch.from_numpy(ops.transform_map(H, W, hfov, vmax, vmin)).to(dev)
    xyz, offs = synth.make_batch(range(B), H, W, device=dev)
    rng = np.random.default_rng(0)
    gms = torch.from_numpy(np.tile(np.array([0., 0., -1., -1.73]), (B, 1)) + rng.normal(0, 0.004, (B, 4))).to(dev)
    ri = ops.project(xyz, offs, geom)
    print("project      %.3f ms" % timeit(lambda: ops.project(xyz, offs, geom, ri=ri)))
    temp0, info = ops.ground_mask(ri, tm, gms, 0.1)
    print("ground_mask  %.3f ms" % timeit(lambda: ops.ground_mask(ri, tm, gms, 0.1)))
    for M in (2, 10, 50, 100):
        def f():
            t = temp0.clone()
            return ops.fps_range(ri, tm, t, info, M)
        tc = timeit(lambda: temp0.clone())
        print("fps_range M=%3d  %.3f ms (clone %.3f)" % (M, timeit(f) - tc, tc))
    temp = temp0.clone(); cen_pix, centers = ops.fps_range(ri, tm, temp, info, 100)
    print("assign       %.3f ms" % timeit(lambda: ops.assign(ri, tm, gms, centers)))
    seg = ops.assign(ri, tm, gms, centers)
    ws = ops.workspace(B, H*W, 100, dev)
    print("point_model  %.3f ms" % timeit(lambda: ops.point_model(ri, seg, gms, 100, ws=ws)))
    model, counts = ops.point_model(ri, seg, gms, 100, ws=ws)
    print("pred_quant   %.3f ms" % timeit(lambda: ops.predict_quantize(ri, tm, seg, model, 0.04, 100, int16=True, ws=ws)))
    buf = ops.BatchBuffers(B, geom, 100, dev)
    print("fused batch  %.3f ms" % timeit(lambda: ops.compress_batch(xyz, offs, tm, gms, buf)))

if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Soak check at the bench geometry: N synthetic 64x2048 frames through the fused entry (ground fitted inside) against the
CPU oracle frame by frame -- range image, FPS pixels, labels, model rows, quantised integers.  usage: soak_fullsize.py [N] [first_id] [geometry]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ThreadPoolExecutor
import numpy as np, torch
import rpcc_amd  # noqa: F401
from rpcc_amd import ops, synth
from oracle import oracle as orc

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
first = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
gname = sys.argv[3] if len(sys.argv) > 3 else "Velodyne64E_2048"      # a key of oracle.GEOMS
dev = torch.device("cuda:0")
gd = orc.GEOMS[gname]
H, W, M = gd["H"], gd["W"], 100
g = orc.LidarGeom(**gd)
tm = orc.transform_map(g)
geom = ops.make_geom(H, W, g.horizontal_FOV, g.vertical_max, g.vertical_min)
orc.lib()
bad = 0
t0 = time.time()
for c0 in range(0, N, 256):
    ids = list(range(first + c0, first + min(c0 + 256, N)))
    xyz, offs = synth.make_batch(ids, H, W, device=dev, vmax_deg=gd["vmax_deg"], vmin_deg=gd["vmin_deg"])
    B = len(ids)
    buf = ops.BatchBuffers(B, geom, M, dev, max_points=xyz.shape[0])
    gfit = torch.zeros((B, 4), dtype=torch.float64, device=dev)
    ops.compress_batch(xyz, offs, torch.from_numpy(tm).to(dev), gfit, buf, ground_seed=77)
    torch.cuda.synchronize()
    o = offs.cpu().numpy(); x = xyz.cpu().numpy()
    ri, seg, cen, q16, nnz, gf, model = (buf.ri.cpu().numpy(), buf.seg.cpu().numpy(), buf.cen_pix.cpu().numpy(), buf.q16.cpu().numpy(),
                                         buf.nnz.cpu().numpy(), gfit.cpu().numpy(), buf.model.cpu().numpy())

    def check(i):
        f = x[o[i]:o[i + 1]]
        r = orc.project(f, g)
        gm = orc.ground_model(r, tm, seed=77 + i)
        e = orc.compress_frame(f, g, tm, gm)
        n = int(nnz[i])
        ok = (np.array_equal(ri[i].view(np.uint32), r.view(np.uint32)) and np.array_equal(gf[i].view(np.uint64), gm.view(np.uint64))
              and np.array_equal(cen[i], e["fps_pix"]) and np.array_equal(seg[i], e["seg_idx"].astype(np.uint8))
              and n == e["q"].shape[0] and np.array_equal(q16[i, :n], e["q"].astype(np.int16))
              and np.array_equal(model[i, : e["model_param"].shape[0]].view(np.uint32), e["model_param"].astype(np.float32).view(np.uint32)))
        return ok
    with ThreadPoolExecutor(os.cpu_count() or 8) as ex:
        res = list(ex.map(check, range(B)))
    bad += res.count(False)
    print("frames %d..%d: %d mismatching" % (ids[0], ids[-1], res.count(False)), flush=True)
print("soak %s: %d frames, %d mismatching, %.0f s" % (gname, N, bad, time.time() - t0))
sys.exit(1 if bad else 0)

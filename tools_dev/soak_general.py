#!/usr/bin/env python3
"""Soak check of the stage-by-stage batch path at the bench geometry: N synthetic 64x2048 frames, non-uniform framework +
plane model, against the CPU oracle frame by frame (labels, plane / mean rows, key points via the quantised integers, salience
levels).  usage: soak_general.py [N] [first_id]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ThreadPoolExecutor
import numpy as np, torch
import rpcc_amd  # noqa: F401
from rpcc_amd import synth
from rpcc_amd.transformer import PCTransformer
from rpcc_amd.pipeline import BatchCompressor
from oracle import oracle as orc

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
first = int(sys.argv[2]) if len(sys.argv) > 2 else 300000
dev = torch.device("cuda:0")
H, W, M = 64, 2048, 100
T = PCTransformer(dict(HORIZONTAL_FOV=360, VERTICAL_ANGLE_MAX=2.0, VERTICAL_ANGLE_MIN=-24.9, RANGE_IMAGE_HEIGHT=H, RANGE_IMAGE_WIDTH=W))
g = orc.LidarGeom(H=H, W=W, hfov_deg=360, vmax_deg=2.0, vmin_deg=-24.9)
tm = orc.transform_map(g)
orc.lib()
cfg = dict(orc.DEFAULT_CFG, plane_angle_threshold=75)
bc = BatchCompressor(T, uniform=False, model_method="plane", compressor_cfg=cfg, seed=9, device=dev)
lacc = np.array([0.04] * 4) + np.array([0, 0.02, 0.04, 0.06])
bad = 0
t0 = time.time()
CH = 128
for c0 in range(0, N, CH):
    ids = list(range(first + c0, first + min(c0 + CH, N)))
    xyz, offs = synth.make_batch(ids, H, W, device=dev)
    buf, gfit, bits, seq, nseq, sal = bc.compress_device(xyz, offs)
    torch.cuda.synchronize()
    o = offs.cpu().numpy(); x = xyz.cpu().numpy()
    seg, q16, nnz, model, salh, gf = (buf.seg.cpu().numpy(), buf.q16.cpu().numpy(), buf.nnz.cpu().numpy(), buf.model.cpu().numpy(),
                                      sal.cpu().numpy(), gfit.cpu().numpy())

    def check(i):
        f = x[o[i]:o[i + 1]]
        ri = orc.project(f, g)
        gm = orc.ground_model(ri, tm, seed=9 + i)
        s = orc.segment(ri, tm, gm, cfg)
        sg = s["seg_idx"]
        mp = np.concatenate((gm.reshape(1, 4), orc.cluster_modeling_plane(s["pc"], ri, sg, tm, 75, 9, i)), 0)
        pred = orc.intra_predict(sg, mp.astype(np.float32), tm)
        _, kp = orc.extract_features_with_segment(ri, sg)
        q, so = orc.nonuniform_quantize(sg, ri.reshape(H, W, 1) - pred, kp, np.array([30, 10, 3, 0]), lacc, 2)
        n = int(nnz[i])
        return (np.array_equal(gf[i].view(np.uint64), gm.view(np.uint64)) and np.array_equal(seg[i], sg.astype(np.uint8))
                and np.array_equal(model[i, : mp.shape[0]].view(np.uint32), mp.astype(np.float32).view(np.uint32))
                and np.array_equal(salh[i, : so.shape[0]], so.astype(np.uint8))
                and n == q.shape[0] and np.array_equal(q16[i, :n], q.astype(np.int16)))
    with ThreadPoolExecutor(os.cpu_count() or 8) as ex:
        res = list(ex.map(check, range(len(ids))))
    bad += res.count(False)
    print("frames %d..%d: %d mismatching" % (ids[0], ids[-1], res.count(False)), flush=True)
print("soak (non-uniform + plane): %d frames, %d mismatching, %.0f s" % (N, bad, time.time() - t0))
sys.exit(1 if bad else 0)

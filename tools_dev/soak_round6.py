#!/usr/bin/env python3
"""Soak of round 6's new paths on the GPU box: (1) the window projection (FORCE and probe) against the record kernels on synthetic sweeps in shuffled,
ring and reversed-ring order, four geometries; (2) cluster_num = 300 and 700 through the fused plan on uint16 labels against the oracle, frame by
frame, both frameworks and models; (3) batches with ground-less sweeps: fused planes against rpcc_ground_ransac alone.
usage: soak_round6.py [frames for (1)] [frames for (2)] [batches for (3)]"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rpcc_amd  # noqa: E402,F401
from rpcc_amd import ops, synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

N1 = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
N2 = int(sys.argv[2]) if len(sys.argv) > 2 else 512
N3 = int(sys.argv[3]) if len(sys.argv) > 3 else 16
dev = torch.device("cuda:0")
t0 = time.time()


def ring_order(f, H):
    el = np.arctan2(f[:, 2], np.hypot(f[:, 0], f[:, 1]))
    ring = np.round((el - el.min()) / (el.max() - el.min() + 1e-9) * (H - 1)).astype(np.int64)
    return f[np.lexsort((np.arctan2(f[:, 1], f[:, 0]), -ring))]


# ---- (1) projection
bad1 = taken = 0
names = ["Velodyne64E_2048", "Velodyne64E", "VelodyneVLP16", "KITTI_like_80x2000"]
geoms = {"KITTI_like_80x2000": dict(H=80, W=2000, hfov_deg=360, vmax_deg=3.0, vmin_deg=-25.0)}
rng = np.random.default_rng(7)
for c0 in range(0, N1, 64):
    gname = names[(c0 // 64) % len(names)]
    gd = geoms.get(gname) or orc.GEOMS[gname]
    g = orc.LidarGeom(**gd)
    geom = ops.make_geom(g.H, g.W, g.horizontal_FOV, g.vertical_max, g.vertical_min)
    frames = []
    for i in range(64):
        f = synth.make_frame(500000 + c0 + i, g.H, g.W, vmax_deg=gd["vmax_deg"], vmin_deg=gd["vmin_deg"]).numpy()
        k = (c0 + i) % 4
        if k == 1:
            f = ring_order(f, g.H)
        elif k == 2:
            f = ring_order(f, g.H)[::-1].copy()
        elif k == 3:
            f = ring_order(f, g.H)
            m = rng.integers(0, f.shape[0], 200)
            f = np.concatenate([f, f[m] * np.float32(0.8)])      # late stragglers: rows re-opened
        frames.append(f)
    offs = np.zeros(65, np.int64)
    offs[1:] = np.cumsum([f.shape[0] for f in frames])
    xyz, o = torch.from_numpy(np.concatenate(frames)).to(dev), torch.from_numpy(offs).to(dev)
    ref = ops.project(xyz, o, geom)
    acc = torch.zeros(64, dtype=torch.int32, device=dev)
    for flags in (ops.PROJECT_ORDER_PROBE, ops.PROJECT_FORCE_ORDERED):
        got = ops.project(xyz, o, geom, order_flags=flags, accepted=acc)
        bad1 += int((got.view(torch.int32) != ref.view(torch.int32)).sum().item())
        if flags == ops.PROJECT_ORDER_PROBE:
            taken += int(acc.sum().item())
print("(1) window projection: %d frames x 2 modes, %d taken by the probe, %d differing pixels, %.0f s" % (N1, taken, bad1, time.time() - t0))

# ---- (2) uint16 labels on the tuned kernels
t1 = time.time()
bad2 = 0
gd = orc.GEOMS["VelodyneVLP16"]
g = orc.LidarGeom(**gd)
geom = ops.make_geom(g.H, g.W, g.horizontal_FOV, g.vertical_max, g.vertical_min)
tm = orc.transform_map(g)
d_tm = torch.from_numpy(tm).to(dev)
orc.lib()
combos = [(300, True, "point"), (300, False, "plane"), (700, True, "plane"), (700, False, "point")]
for c0 in range(0, N2, 32):
    M, uniform, method = combos[(c0 // 32) % len(combos)]
    ids = list(range(600000 + c0, 600000 + c0 + 32))
    xyz, offs = synth.make_batch(ids, g.H, g.W, device=dev, vmax_deg=gd["vmax_deg"], vmin_deg=gd["vmin_deg"])
    buf = ops.BatchBuffers(32, geom, M, dev)
    gms = torch.zeros((32, 4), dtype=torch.float64, device=dev)
    cfg = dict(orc.DEFAULT_CFG, cluster_num=M, plane_angle_threshold=75)
    nu = None if uniform else ops.nonuniform_cfg(0.04, cfg)
    fid = torch.as_tensor(np.asarray(ids, np.int64), device=dev)
    ops.compress_batch(xyz, offs, d_tm, gms, buf, ground_seed=11, frame_ids=fid, model_method=method, plane_seed=11, nonuniform=nu)
    torch.cuda.synchronize()
    o_h, x_h = offs.cpu().numpy(), xyz.cpu().numpy()
    seg, q16, nnz, gm_d, sal = buf.seg.cpu().numpy(), buf.q16.cpu().numpy(), buf.nnz.cpu().numpy(), gms.cpu().numpy(), buf.salience.cpu().numpy()

    def check(i):
        f = x_h[o_h[i]:o_h[i + 1]]
        gm = orc.ground_model(orc.project(f, g), tm, seed=11 + ids[i])
        o = orc.compress_frame(f, g, tm, gm, cfg, uniform=uniform, plane=None if method == "point" else dict(angle_deg=75, seed=11, frame=ids[i]))
        ok = np.array_equal(gm_d[i].view(np.uint64), np.asarray(gm, np.float64).view(np.uint64)) and \
            np.array_equal(seg[i].reshape(-1), o["seg_idx"].reshape(-1)) and int(nnz[i]) == o["q"].shape[0] and \
            np.array_equal(q16[i, :nnz[i]], o["q"].astype(np.int16))
        if ok and not uniform:
            ok = np.array_equal(sal[i, :o["salience"].shape[0]], o["salience"].astype(np.uint8))
        return 0 if ok else 1
    with ThreadPoolExecutor(16) as ex:
        bad2 += sum(ex.map(check, range(32)))
print("(2) 300 / 700 clusters on uint16 labels: %d frames, %d mismatching, %.0f s" % (N2, bad2, time.time() - t1))

# ---- (3) ground-less sweeps
t2 = time.time()
bad3 = 0
gd = orc.GEOMS["Velodyne64E_2048"]
g = orc.LidarGeom(**gd)
geom = ops.make_geom(g.H, g.W, g.horizontal_FOV, g.vertical_max, g.vertical_min)
d_tm = torch.from_numpy(orc.transform_map(g)).to(dev)
for r in range(N3):
    ids = list(range(700000 + 64 * r, 700000 + 64 * r + 64))
    frames = [synth.make_frame(i, g.H, g.W).numpy() for i in ids]
    pick = rng.random(64) < 0.3
    frames = [f[f[:, 2] > -1.45] if p else f for f, p in zip(frames, pick)]
    offs = np.zeros(65, np.int64)
    offs[1:] = np.cumsum([f.shape[0] for f in frames])
    buf = ops.BatchBuffers(64, geom, 100, dev)
    gms = torch.zeros((64, 4), dtype=torch.float64, device=dev)
    fid = torch.as_tensor(np.asarray(ids, np.int64), device=dev)
    ops.compress_batch(torch.from_numpy(np.concatenate(frames)).to(dev), torch.from_numpy(offs).to(dev), d_tm, gms, buf, ground_seed=4, frame_ids=fid)
    alone, _ = ops.ground_ransac(buf.ri, d_tm, seed=4, frame_ids=fid)
    bad3 += int((gms.view(torch.int64) != alone.view(torch.int64)).any(1).sum().item())
print("(3) ground-less sweeps: %d batches of 64 (30 %% without ground), %d planes differing from the fit alone, %.0f s" % (N3, bad3, time.time() - t2))
sys.exit(1 if (bad1 or bad2 or bad3) else 0)

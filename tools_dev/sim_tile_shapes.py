"""CPU study (statistics only): how many FPS tiles a new centre touches per iteration for a lidar geometry under different tile shapes
(rows x columns; the shipped kernels use 8 x 32 = 256 pixels, four per lane).  Prints, per shape, the mean number of tiles whose bound
test passes per iteration and the mean of the busiest wavefront's share (tiles are dealt to the eight wavefronts round-robin in the
row-rotated order the kernel uses; the first pass, the mask kernel's, is left out).
Usage: python tools_dev/sim_tile_shapes.py <lidar> [frame ids...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpcc_amd  # noqa: E402,F401
from rpcc_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

M = 100
f32 = np.float32


def run(lidar, fid, shapes):
    gd = orc.GEOMS[lidar]
    g = orc.LidarGeom(**gd)
    H, W = g.H, g.W
    tm = orc.transform_map(g)
    xyz = synth.make_frame(fid, H, W, vmax_deg=gd["vmax_deg"], vmin_deg=gd["vmin_deg"]).numpy()
    ri = orc.project(xyz, g)
    gm = orc.ground_model(ri, tm, seed=fid)
    o = orc.compress_frame(xyz, g, tm, gm)
    pc = orc.backproject(ri, tm).reshape(-1, 3).astype(f32)
    P = H * W
    rif = ri.reshape(-1)
    cand = o["mask"].reshape(-1) & (rif != 0)
    cen = o["centers"].astype(f32)
    rows, cols = np.arange(P) // W, np.arange(P) % W
    out = []
    for TR, TC in shapes:
        tcols = (W + TC - 1) // TC
        trows = (H + TR - 1) // TR
        tile = (rows // TR) * tcols + cols // TC
        T = trows * tcols
        # boxes of the candidates
        lo = np.full((T, 3), np.inf, f32); hi = np.full((T, 3), -np.inf, f32)
        ci = np.nonzero(cand)[0]
        np.minimum.at(lo, tile[ci], pc[ci]); np.maximum.at(hi, tile[ci], pc[ci])
        has = np.isfinite(lo[:, 0])
        # position of a tile in the kernel's order: pos = tr * tcols + (tc + 3 tr) % tcols
        tr_, tc_ = np.arange(T) // tcols, np.arange(T) % tcols
        pos = tr_ * tcols + (tc_ + 3 * tr_) % tcols
        nw = 8
        wave = pos % nw                                # the register table deals positions round-robin
        temp = np.full(P, f32(1e10))
        tmax = np.where(has, f32(1e10), f32(-1))
        touched, busiest, waves_busy = [], [], []
        for k in range(M - 1):
            c = cen[k]
            gap = np.maximum(np.maximum(lo - c, c - hi), 0).astype(f32)
            bound = ((gap[:, 0] * gap[:, 0] + gap[:, 1] * gap[:, 1]) + gap[:, 2] * gap[:, 2]).astype(f32)
            vis = has & (bound < tmax)
            if k > 0:                                  # the first pass belongs to the mask kernel
                touched.append(int(vis.sum()))
                cnt = np.bincount(wave[vis], minlength=nw)
                busiest.append(int(cnt.max())); waves_busy.append(int((cnt > 0).sum()))
            sel = vis[tile] & cand
            d = pc[sel] - c
            d2 = ((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]).astype(f32)
            temp[sel] = np.minimum(temp[sel], d2)
            tm_new = np.full(T, f32(-1)); np.maximum.at(tm_new, tile[ci], temp[ci])
            tmax = np.where(has, tm_new, f32(-1))
        out.append((TR, TC, T, int(has.sum()), np.mean(touched), np.mean(busiest), np.mean(waves_busy), np.mean(touched) * TR * TC))
    return out


if __name__ == "__main__":
    lidar = sys.argv[1] if len(sys.argv) > 1 else "Velodyne64E"
    fids = [int(a) for a in sys.argv[2:]] or [0, 1, 2]
    shapes = [(8, 32), (4, 64), (16, 16), (4, 32), (8, 16), (2, 64), (2, 32), (4, 16), (1, 64)]
    acc = {}
    for f in fids:
        for r in run(lidar, f, shapes):
            acc.setdefault(r[:2], []).append(r[2:])
    print("%s: mean over frames %s and the 99 iterations" % (lidar, fids))
    print("rows x cols | tiles | with candidates | tiles touched / iteration | busiest wavefront's tiles | wavefronts with work | pixels touched / iteration")
    for (tr, tc), v in acc.items():
        a = np.mean(np.array(v, float), axis=0)
        print("%2d x %3d | %5d | %5d | %6.1f | %5.2f | %4.2f | %7.0f" % (tr, tc, a[0], a[1], a[2], a[3], a[4], a[5]))

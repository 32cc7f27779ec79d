"""CPU study (statistics only): which centres ever lower temp inside an FPS tile (the per-tile centre masks the FPS could hand to the
assignment), against the number of distinct nearest centres of the tile's candidates and the survivors of assign_kernel's reach screen.
Usage: python tools_dev/sim_tile_masks.py [frame ids...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpcc_amd  # noqa: E402,F401
from rpcc_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

H, W, M = 64, 2048, 100
TR, TC = 8, 32


def run(fid):
    g = orc.LidarGeom(H=H, W=W)
    tm = orc.transform_map(g)
    xyz = synth.make_frame(fid, H, W).numpy()
    ri = orc.project(xyz, g)
    gm = orc.ground_model(ri, tm, seed=fid)
    o = orc.compress_frame(xyz, g, tm, gm)
    pc = orc.backproject(ri, tm).reshape(-1, 3).astype(np.float32)
    mask = o["mask"].reshape(-1)
    P = H * W
    rif = ri.reshape(-1)
    cand = mask & (rif != 0)
    tcols = W // TC
    tile = ((np.arange(P) // W) // TR) * tcols + (np.arange(P) % W) // TC
    T = int(tile.max()) + 1
    cen = o["centers"].astype(np.float32)
    temp = np.full(P, np.float32(1e10))
    near = np.zeros(P, np.int32)
    tmask = np.zeros((T, M), bool)
    for k in range(M):
        d = pc - cen[k]
        d2 = ((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]).astype(np.float32)
        ch = cand & (d2 < temp)
        temp = np.where(ch, d2, temp)
        near[ch] = k
        tmask[np.unique(tile[ch]), k] = True
    has = np.array([cand[tile == t].any() for t in range(T)])
    bits = tmask.sum(1)
    distinct = np.array([len(np.unique(near[(tile == t) & cand])) if has[t] else 0 for t in range(T)])
    allc = np.array([(cand | (rif == 0))[tile == t].all() for t in range(T)])   # tiles without a ground pixel
    print("frame %d: tiles %d, with candidates %d, candidates only %d | mask bits per tile with candidates: mean %.2f max %d | "
          "distinct nearest: mean %.2f | candidates-only tiles: bits %.2f distinct %.2f" %
          (fid, T, has.sum(), (allc & has).sum(), bits[has].mean(), bits.max(), distinct[has].mean(),
           bits[allc & has].mean(), distinct[allc & has].mean()))


if __name__ == "__main__":
    for f in ([int(a) for a in sys.argv[1:]] or [0, 1, 2]):
        run(f)

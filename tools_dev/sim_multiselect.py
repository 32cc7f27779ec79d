"""CPU simulation (statistics only): how many FPS centres can be committed per dependent iteration when the next few
centres are taken from the per-tile maxima under a conservative validity test (DESIGN.md "FPS: multi-select")?
Usage: python tools_dev/sim_multiselect.py [frame ids...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpcc_amd  # noqa: E402,F401
from rpcc_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

H, W, M = 64, 2048, 100
TR, TC = 8, 32
KMAX = int(os.environ.get("KMAX", "4"))


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
    tile = ((np.arange(P) // W) // TR) * (W // TC) + (np.arange(P) % W) // TC
    T = tile.max() + 1
    order = np.argsort(tile, kind="stable")
    temp = np.where(mask, np.float32(1e10), np.float32(-1))
    first = int(np.flatnonzero(mask)[0])
    # boxes of the candidates per tile
    lo = np.full((T, 3), np.inf, np.float32); hi = np.full((T, 3), -np.inf, np.float32)
    for a in range(3):
        np.minimum.at(lo[:, a], tile[mask], pc[mask, a]); np.maximum.at(hi[:, a], tile[mask], pc[mask, a])
    sel = [first]

    def apply(c):
        d = ((pc - pc[c]) ** 2).sum(1).astype(np.float32)
        upd = mask & (d < temp)
        temp[upd] = d[upd]
        return len(np.unique(tile[upd]))

    apply(first)
    iters = 0
    visits = 0
    hist = np.zeros(KMAX + 1, int)
    while len(sel) < M:
        iters += 1
        # tile maxima
        tmax = np.full(T, -1.0, np.float32); np.maximum.at(tmax, tile, temp)
        targ = np.full(T, P, np.int64)
        isbest = temp == tmax[tile]
        np.minimum.at(targ, tile[isbest], np.arange(P)[isbest])
        ranking = np.lexsort((targ, -tmax))           # value desc, index asc
        chosen = [int(targ[ranking[0]])]
        ctiles = [int(ranking[0])]
        for r in range(1, KMAX):
            if len(sel) + len(chosen) >= M:
                break
            t2 = int(ranking[r]); c = int(targ[t2]); tv = tmax[t2]
            ok = tv > 0
            for cm, tmm in zip(chosen, ctiles):
                d = ((pc[c] - pc[cm]) ** 2).sum()
                if d < tv:
                    ok = False
                # points of the earlier centre's tile are bounded by the farthest box distance
                f = np.maximum(np.abs(lo[tmm] - pc[cm]), np.abs(hi[tmm] - pc[cm]))
                if not ((f ** 2).sum() < tv):
                    ok = False
            if not ok:
                break
            chosen.append(c); ctiles.append(t2)
        hist[len(chosen)] += 1
        for c in chosen:
            visits += apply(c)
            sel.append(c)
    assert np.array_equal(np.array(sel[:M]), o["fps_pix"][:M]), "multi-select changed the sequence"
    return iters, hist, visits


if __name__ == "__main__":
    ids = [int(a) for a in sys.argv[1:]] or [0, 1, 2]
    for fid in ids:
        it, hist, visits = run(fid)
        print("frame %d: %d dependent iterations for %d centres; group sizes %s; tiles changed %d" % (fid, it, M - 1, hist[1:], visits))

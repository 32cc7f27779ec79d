"""CPU simulation (statistics only) of the LAZY tile rule of the FPS kernel: a tile whose box a new
centre can reach is not visited at once -- the centre is queued on the tile, the tile's maximum becomes an upper bound
(tightened by the farthest-corner distance of the centre to the tile's box) -- and a stale tile is brought up to date only when
its bound reaches a lower bound L of the next global maximum that every wavefront can compute by itself from the candidates
the wavefronts published in the previous iteration: L = max_k min(v_k, d(c, p_k)).
Prints, per frame: visits of the eager rule (the shipped kernel), refreshes of the lazy rule, iterations without any refresh,
the most loaded wavefront's refreshes per iteration.   Usage: python tools_dev/sim_lazy_fps.py [frame ids...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpcc_amd  # noqa: E402,F401
from rpcc_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

H, W, M = 64, 2048, 100
TR, TC, NW = 8, 32, 8
MAXPEND = int(os.environ.get("MAXPEND", "1000"))
USE_MAXDIST = os.environ.get("MAXDIST", "1") == "1"


def run(fid, real=None):
    g = orc.LidarGeom(H=H, W=W)
    tm = orc.transform_map(g)
    xyz = synth.make_frame(fid, H, W).numpy() if real is None else real
    ri = orc.project(xyz, g)
    gm = orc.ground_model(ri, tm, seed=fid)
    o = orc.compress_frame(xyz, g, tm, gm)
    pc = orc.backproject(ri, tm).reshape(-1, 3).astype(np.float32)
    mask = o["mask"].reshape(-1)
    P = H * W
    rif = ri.reshape(-1)
    org = mask & (rif == 0)          # origin class
    cand = mask & ~org
    tcols = W // TC
    tr = (np.arange(P) // W) // TR
    tc = (np.arange(P) % W) // TC
    tile = tr * tcols + tc
    T = int(tile.max()) + 1
    pos = (np.arange(T) // tcols) * tcols + ((np.arange(T) % tcols) + 3 * (np.arange(T) // tcols)) % tcols
    wave_of = pos % NW
    pts = [np.flatnonzero((tile == t) & cand) for t in range(T)]
    lo = np.full((T, 3), np.inf, np.float32); hi = np.full((T, 3), -np.inf, np.float32)
    for t in range(T):
        if len(pts[t]):
            lo[t] = pc[pts[t]].min(0); hi[t] = pc[pts[t]].max(0)
    temp = np.where(mask, np.float32(1e10), np.float32(-1))
    first = int(np.flatnonzero(mask)[0])
    t_org = np.float32(1e10) if org.any() else np.float32(-1)
    org_idx = int(np.flatnonzero(org)[0]) if org.any() else P

    def dist(c, idx):
        d = pc[idx] - c
        return ((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]).astype(np.float32)

    def exact_max(t):
        if not len(pts[t]):
            return np.float32(-1), P
        v = temp[pts[t]]
        m = v.max()
        return m, int(pts[t][np.argmax(v == m)])

    # first centre applied eagerly everywhere (the ground-mask kernel's first pass)
    c = pc[first]
    for t in range(T):
        if len(pts[t]):
            temp[pts[t]] = np.minimum(temp[pts[t]], dist(c, pts[t]))
    if org.any():
        t_org = min(t_org, np.float32(((pc[org_idx] - c) ** 2).sum()))
    tmax = np.zeros(T, np.float32); targ = np.zeros(T, np.int64)
    for t in range(T):
        tmax[t], targ[t] = exact_max(t)
    ub = tmax.copy()
    fresh = np.ones(T, bool)
    pend = [[] for _ in range(T)]
    has = np.array([len(p) > 0 for p in pts])
    sel = [first]
    eager_visits = 0
    eager_temp = temp.copy()
    eager_tmax = tmax.copy()
    stats = dict(refresh=0, zero_iters=0, maxwave=0, rounds=0, pend_hist=np.zeros(64, int), forced=0, eager_maxwave=0, eager_zero=0)
    cands = []     # published candidates of the previous iteration: (value, point index)

    def select():
        best_v, best_i = np.float32(-1), P
        pub = []
        for w in range(NW):
            m = fresh & has & (wave_of == w)
            if m.any():
                v = tmax[m].max()
                i = int(targ[m][tmax[m] == v].min())
                pub.append((v, i))
                if v > best_v or (v == best_v and i < best_i):
                    best_v, best_i = v, i
        if t_org >= 0 and (t_org > best_v or (t_org == best_v and org_idx < best_i)):
            best_v, best_i = t_org, org_idx
        return best_i, pub

    nxt, cands = select()
    sel.append(nxt)
    for j in range(2, M):
        c = pc[sel[-1]]
        # eager rule (shipped kernel) bookkeeping, on its own copy
        g0 = np.maximum(np.maximum(lo - c, c - hi), 0)
        lb = ((g0[:, 0] * g0[:, 0] + g0[:, 1] * g0[:, 1]) + g0[:, 2] * g0[:, 2]).astype(np.float32)
        ev = has & (lb < eager_tmax)
        eager_visits += int(ev.sum())
        stats["eager_maxwave"] += int(np.bincount(wave_of[ev], minlength=NW).max()) if ev.any() else 0
        stats["eager_zero"] += int(not ev.any())
        for t in np.flatnonzero(ev):
            eager_temp[pts[t]] = np.minimum(eager_temp[pts[t]], dist(c, pts[t]))
            eager_tmax[t] = eager_temp[pts[t]].max()
        # lazy rule
        f = np.maximum(np.abs(lo - c), np.abs(hi - c))
        md = ((f[:, 0] * f[:, 0] + f[:, 1] * f[:, 1]) + f[:, 2] * f[:, 2]).astype(np.float32)
        aff = has & (lb < ub)
        for t in np.flatnonzero(aff):
            pend[t].append(sel[-1])
            fresh[t] = False
            if USE_MAXDIST:
                ub[t] = min(ub[t], md[t])
        if t_org >= 0:
            t_org = min(t_org, np.float32(((pc[org_idx] - c) ** 2).sum()))
        # lower bound of the new global maximum from last iteration's published candidates
        L = np.float32(-1)
        for v, i in cands:
            d = np.float32(((pc[i] - c) ** 2).sum()) if i != sel[-1] else np.float32(0)
            L = max(L, min(v, d))
        if t_org >= 0:
            L = max(L, t_org)
        need = (~fresh) & has & ((ub >= L) | np.array([len(p) > MAXPEND for p in pend]))
        stats["forced"] += int(((~fresh) & has & ~(ub >= L) & np.array([len(p) > MAXPEND for p in pend])).sum())
        nref = int(need.sum())
        stats["refresh"] += nref
        stats["zero_iters"] += int(nref == 0)
        if nref:
            stats["maxwave"] += int(np.bincount(wave_of[need], minlength=NW).max())
        for t in np.flatnonzero(need):
            stats["pend_hist"][min(len(pend[t]), 63)] += 1
            for ci in pend[t]:
                temp[pts[t]] = np.minimum(temp[pts[t]], dist(pc[ci], pts[t]))
            pend[t] = []
            tmax[t], targ[t] = exact_max(t)
            ub[t] = tmax[t]
            fresh[t] = True
        nxt, cands = select()
        sel.append(nxt)
    ok = np.array_equal(np.array(sel[:M]), o["fps_pix"][:M])
    return ok, eager_visits, stats


if __name__ == "__main__":
    ids = [int(a) for a in sys.argv[1:]] or [0, 1, 2]
    for fid in ids:
        ok, ev, st = run(fid)
        n = M - 2
        print("frame %d: sequence %s | eager: %.2f visits/iter, busiest wave %.2f, %d iterations without a visit | lazy: %.2f refreshes/iter, "
              "busiest wave %.2f, %d of %d iterations without a refresh, forced %d, pending centres per refresh: mean %.2f max %d"
              % (fid, "OK" if ok else "DIFFERS", ev / n, st["eager_maxwave"] / n, st["eager_zero"], st["refresh"] / n, st["maxwave"] / n, st["zero_iters"], n,
                 st["forced"], (st["pend_hist"] * np.arange(64)).sum() / max(st["pend_hist"].sum(), 1), int(np.flatnonzero(st["pend_hist"]).max()) if st["pend_hist"].any() else 0))

"""(CPU) Simulation of the record-free projection's sliding LDS window (project_ordered_kernel) on a sweep in stored order: how often the
window moves, how many rows are retired / re-opened, passes per chunk.  usage: python tools_dev/sim_ordered_window.py [npz] [chunk] [--shuffle]"""
import sys
import numpy as np

sys.path.insert(0, ".")
from oracle import oracle as orc   # noqa: E402

path = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "tests/golden/example_64E.npz"
chunk = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else 8192
xyz = np.load(path)["xyz"]
if "--shuffle" in sys.argv:
    xyz = xyz[np.random.default_rng(0).permutation(len(xyz))]
H, W = 64, 2000
gd = orc.GEOMS["Velodyne64E"]
x, y, z = xyz.T.astype(np.float64)
az = np.arctan2(y, x); az[az < 0] += 2 * np.pi
el = np.arctan2(z, np.hypot(x, y))
vmax, vmin = np.deg2rad(gd["vmax_deg"]), np.deg2rad(gd["vmin_deg"])
row = np.clip(np.round((el - vmin) / ((vmax - vmin) / (H - 1))), 0, H - 1).astype(int)
WR = 32768 // W
lo = None
retired = np.zeros(H, bool)
moves = retire_rows = reopen_rows = passes = 0
for c0 in range(0, len(row), chunk):
    r = row[c0:c0 + chunk]
    pend = np.ones(len(r), bool)
    while pend.any():
        passes += 1
        if lo is not None:
            inw = (r >= lo) & (r < lo + WR)
            pend &= ~inw
            if not pend.any():
                break
        hist = np.bincount(r[pend], minlength=H)
        cover = np.array([hist[l:l + WR].sum() for l in range(0, H - WR + 1)])
        best = cover.max()
        cands = np.flatnonzero(cover == best)
        new = cands[np.argmin(np.abs(cands - (lo if lo is not None else cands[0])))]
        if lo is None:
            lo = new
            continue
        old_rows = set(range(lo, lo + WR)); new_rows = set(range(new, new + WR))
        for rr in old_rows - new_rows:
            retired[rr] = True; retire_rows += 1
        for rr in new_rows - old_rows:
            if retired[rr]:
                reopen_rows += 1
        moves += 1
        lo = new
nchunks = (len(row) + chunk - 1) // chunk
print("points %d chunks %d WR %d: moves %d, rows retired %d, rows re-opened %d, passes %d (%.2f per chunk)" %
      (len(row), nchunks, WR, moves, retire_rows, reopen_rows, passes, passes / nchunks))

# ---- the same with 3 % of the points (the exact-sequence queue) applied one chunk late
rng = np.random.default_rng(0)
late = rng.random(len(row)) < 0.03
lo = None; retired[:] = False; moves = retire_rows = reopen_rows = 0
prev_late = np.zeros(0, int)
for c0 in range(0, len(row) + chunk, chunk):
    idx = np.arange(c0, min(c0 + chunk, len(row)))
    r = np.concatenate([row[idx][~late[idx]], prev_late])
    prev_late = row[idx][late[idx]]
    pend = np.ones(len(r), bool)
    while pend.any():
        if lo is not None:
            pend &= ~((r >= lo) & (r < lo + WR))
            if not pend.any():
                break
        hist = np.bincount(r[pend], minlength=H)
        cover = np.array([hist[l:l + WR].sum() for l in range(0, H - WR + 1)])
        cands = np.flatnonzero(cover == cover.max())
        new = cands[np.argmin(np.abs(cands - (lo if lo is not None else cands[0])))]
        if lo is None:
            lo = new
            continue
        old_rows = set(range(lo, lo + WR)); new_rows = set(range(new, new + WR))
        retire_rows += len(old_rows - new_rows)
        for rr in old_rows - new_rows:
            retired[rr] = True
        reopen_rows += sum(1 for rr in new_rows - old_rows if retired[rr])
        moves += 1
        lo = new
print("with the uncertain 3 %% one chunk late: moves %d, rows retired %d, rows re-opened %d" % (moves, retire_rows, reopen_rows))

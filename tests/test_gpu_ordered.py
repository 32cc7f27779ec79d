"""-m gpu: a2 for sweeps in scanner order -- project_ordered_kernel (csrc/project_ordered.h: a window of image rows in LDS, no per-point
records, chosen per frame by a probe of the points' order) against the oracle (cpp_modules.cpp:427-467 restated) and against the two
record kernels, bit for bit, for ANY order of the points: the stored order of a real sweep (dataset/dataset.py:48-50), reversed, shuffled,
rings in random order, late points that make the window re-open rows it has written, depth-0 points, special values, ragged batches that
mix accepted and rejected frames, both point layouts, and the fused batch's hand-over to the ground fit."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
PROBE, FORCE = 16, 32      # include/rpcc_hip.h: RPCC_PROJECT_ORDER_PROBE, RPCC_PROJECT_FORCE_ORDERED


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    import rpcc_amd  # noqa: F401
    from rpcc_amd import ops, synth
    from oracle import oracle as orc
    return dict(torch=torch, ops=ops, synth=synth, orc=orc, dev=torch.device("cuda:0"))


def _geom(env, name):
    orc, ops = env["orc"], env["ops"]
    g = orc.LidarGeom(**orc.GEOMS[name])
    tm = ops.transform_map(g.H, g.W, g.horizontal_FOV, g.vertical_max, g.vertical_min)
    return g, ops.make_geom(g.H, g.W, g.horizontal_FOV, g.vertical_max, g.vertical_min), tm


def _to(env, a):
    return env["torch"].from_numpy(np.ascontiguousarray(a)).to(env["dev"])


def _beq(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8))


def _project(env, frames, geom, flags, rows=False):
    torch, ops = env["torch"], env["ops"]
    offs = np.zeros(len(frames) + 1, np.int64)
    offs[1:] = np.cumsum([f.shape[0] for f in frames])
    xyz = np.concatenate(frames) if offs[-1] else np.zeros((0, 3), np.float32)
    if rows:
        r4 = np.full((xyz.shape[0], 4), np.nan, np.float32)
        r4[:, :3] = xyz
        xyz = r4
    acc = torch.full((len(frames),), -1, dtype=torch.int32, device=env["dev"])
    xt = _to(env, xyz) if offs[-1] else torch.zeros((0, 4 if rows else 3), dtype=torch.float32, device=env["dev"])
    ri = ops.project(xt, _to(env, offs), geom, order_flags=flags, accepted=acc)
    return ri.cpu().numpy(), acc.cpu().numpy()


def _ring_order(f, H=64):
    """a synthetic sweep re-ordered ring by ring (coarse elevation bins top down, azimuth inside a ring)"""
    el = np.arctan2(f[:, 2], np.hypot(f[:, 0], f[:, 1]))
    ring = np.round((el - el.min()) / (el.max() - el.min() + 1e-9) * (H - 1)).astype(np.int64)
    return f[np.lexsort((np.arctan2(f[:, 1], f[:, 0]), -ring))]


def test_stored_order_is_accepted_and_equals_the_oracle(env):
    """The real sweep as stored (122 k points, ring by ring) and the orders a probe must tell apart."""
    orc = env["orc"]
    g, geom, _ = _geom(env, "Velodyne64E")
    xyz = np.load(os.path.join(HERE, "golden", "example_64E.npz"))["xyz"]
    rng = np.random.default_rng(5)
    pieces = np.array_split(np.arange(xyz.shape[0]), 64)
    ringshuf = xyz[np.concatenate([pieces[i] for i in rng.permutation(64)])]
    half = xyz.shape[0] // 2
    frames = [xyz, xyz[::-1].copy(), xyz[rng.permutation(xyz.shape[0])], ringshuf, np.concatenate([xyz[half:], xyz[:half]]), xyz[:3000].copy()]
    want = [orc.project(f, g) for f in frames]
    ri, acc = _project(env, frames, geom, PROBE)
    assert list(acc) == [1, 1, 0, 0, 1, 0], acc        # stored, reversed: taken; shuffled, random rings: records; halves swapped: taken; small: records
    for i in range(len(frames)):
        assert _beq(ri[i], want[i]), i
    ri_f, acc_f = _project(env, frames, geom, FORCE)   # every frame through the window kernel, whatever its order
    assert list(acc_f) == [1] * len(frames)
    for i in range(len(frames)):
        assert _beq(ri_f[i], want[i]), ("forced", i)
    ri_n, acc_n = _project(env, frames, geom, 0)
    assert list(acc_n) == [0] * len(frames) and _beq(ri_n, ri)
    ri_r, acc_r = _project(env, frames, geom, PROBE, rows=True)        # the rows as a .bin stores them (16-byte loads, garbage 4th column)
    assert list(acc_r) == list(acc) and _beq(ri_r, ri)
    ri_rf, _ = _project(env, frames, geom, FORCE, rows=True)
    assert _beq(ri_rf, ri)


def test_window_reopens_rows_for_late_points(env):
    """Points that come back to rows the window has already written (a second sweep appended, single stragglers far behind the front,
    a depth-0 point, NaN / inf / huge coordinates, points on the azimuth seam): same image as the oracle's sequential loop."""
    orc, synth = env["orc"], env["synth"]
    gb = orc.LidarGeom(H=64, W=2048, hfov_deg=360, vmax_deg=2.0, vmin_deg=-24.9)
    geomb = env["ops"].make_geom(gb.H, gb.W, gb.horizontal_FOV, gb.vertical_max, gb.vertical_min)
    rng = np.random.default_rng(11)
    f = _ring_order(synth.make_frame(7, 64, 2048).numpy())
    f2 = _ring_order(synth.make_frame(8, 64, 2048).numpy())
    two = np.concatenate([f, f2 * np.float32(0.97)])                   # the second sweep re-opens every row, nearer returns win
    strag = f.copy()
    idx = rng.integers(0, f.shape[0], 400)
    strag[np.sort(rng.integers(f.shape[0] // 2, f.shape[0], 400))] = f[idx] * np.float32(0.5)   # stragglers from anywhere, late
    special = f.copy()
    special[1000] = [np.nan, 1, 1]; special[2000] = [np.inf, 1, 1]; special[3000] = [1e30, 1e30, 0]
    special[4000:4010] = [[5, -1e-9, 0.0]] * 10
    special[5000] = [0, 0, 7]; special[5001] = [0, 0, -7]
    keep = np.ones(f.shape[0], bool); keep[[1000, 2000, 3000]] = False
    zero = f.copy()
    zero[f.shape[0] // 3] = 0                                           # depth-0 point: the frame is redone in input order
    frames = [two, strag, special, zero, f[:5000].copy(), np.zeros((0, 3), np.float32), f[:1].copy()]
    want = [orc.project(two, gb), orc.project(strag, gb), orc.project(special[keep], gb), orc.project(zero, gb), orc.project(f[:5000], gb),
            orc.project(np.zeros((0, 3), np.float32), gb), orc.project(f[:1], gb)]
    for flags in (PROBE, FORCE):
        for rows in (False, True):
            ri, acc = _project(env, frames, geomb, flags, rows=rows)
            for i in range(len(frames)):
                assert _beq(ri[i], want[i]), (flags, rows, i, acc)
            if flags == FORCE:
                assert list(acc) == [1, 1, 1, 1, 1, 0, 1]                # (a frame without a point has nothing to probe)


@pytest.mark.parametrize("gname", ["Velodyne64E", "Velodyne64E_2048", "VelodyneVLP16", "Velodyne32E"])
def test_forced_window_kernel_on_every_shipped_geometry(env, gname):
    """Shuffled synthetic sweeps through the window kernel (FORCE: the window thrashes, the image must not care) on the reference's lidar
    tables: 16 rows x 1800 fits the window whole; 32 x 2250 has a width that is no multiple of four and is never probed."""
    orc, synth = env["orc"], env["synth"]
    gd = orc.GEOMS[gname]
    g, geom, _ = _geom(env, gname)
    frames = [synth.make_frame(300 + i, g.H, g.W, vmax_deg=gd["vmax_deg"], vmin_deg=gd["vmin_deg"]).numpy() for i in range(5)]
    frames[1] = _ring_order(frames[1], g.H)
    frames[3] = frames[3][:1500].copy()
    want = [orc.project(f, g) for f in frames]
    for flags in (PROBE, FORCE):
        ri, acc = _project(env, frames, geom, flags)
        for i in range(len(frames)):
            assert _beq(ri[i], want[i]), (flags, i)
        if g.W % 4:
            assert list(acc) == [0] * 5
        elif flags == FORCE:
            assert list(acc) == [1] * 5
        else:
            assert acc[1] == 1 and acc[3] == 0


def test_fused_batch_mixes_accepted_and_rejected_frames(env):
    """rpcc_compress_batch on a batch that interleaves ring-ordered sweeps (window kernel), shuffled ones (records) and a frame with a
    depth-0 point, ground fitted inside (the candidate counts and bytes the projection hands to the ground fit come from either kernel):
    every output equals the same call without the probe, and the oracle."""
    torch, ops, orc, synth = env["torch"], env["ops"], env["orc"], env["synth"]
    gb = orc.LidarGeom(H=64, W=2048, hfov_deg=360, vmax_deg=2.0, vmin_deg=-24.9)
    geomb = ops.make_geom(gb.H, gb.W, gb.horizontal_FOV, gb.vertical_max, gb.vertical_min)
    tm = ops.transform_map(gb.H, gb.W, gb.horizontal_FOV, gb.vertical_max, gb.vertical_min)
    frames = []
    for i in range(12):
        f = synth.make_frame(900 + i, 64, 2048).numpy()
        frames.append(_ring_order(f) if i % 3 != 1 else f)
    frames[6] = frames[6].copy(); frames[6][777] = 0
    frames[9] = np.concatenate([frames[9], frames[9][:20000] * np.float32(0.9)])     # late points: rows re-opened, candidate counts taken back
    offs = np.zeros(len(frames) + 1, np.int64)
    offs[1:] = np.cumsum([f.shape[0] for f in frames])
    xyz = _to(env, np.concatenate(frames))
    fid = torch.arange(100, 100 + len(frames), dtype=torch.int64, device=env["dev"])
    outs = []
    for flags in (PROBE, 0, FORCE):
        buf = ops.BatchBuffers(len(frames), geomb, 100, env["dev"])
        gms = torch.zeros((len(frames), 4), dtype=torch.float64, device=env["dev"])
        ops.compress_batch(xyz, _to(env, offs), _to(env, tm), gms, buf, ground_seed=3, frame_ids=fid, project_flags=flags)
        torch.cuda.synchronize()
        outs.append(dict(ri=buf.ri.cpu().numpy().copy(), gm=gms.cpu().numpy().copy(), seg=buf.seg.cpu().numpy().copy(), pix=buf.cen_pix.cpu().numpy().copy(),
                         nnz=buf.nnz.cpu().numpy().copy(), q=buf.q16.cpu().numpy().copy(), model=buf.model.cpu().numpy().copy()))
    for o in outs[1:]:
        assert _beq(o["ri"], outs[0]["ri"]) and _beq(o["gm"], outs[0]["gm"]) and np.array_equal(o["seg"], outs[0]["seg"])
        assert np.array_equal(o["pix"], outs[0]["pix"]) and np.array_equal(o["nnz"], outs[0]["nnz"]) and _beq(o["model"], outs[0]["model"])
        assert all(np.array_equal(o["q"][i, :o["nnz"][i]], outs[0]["q"][i, :o["nnz"][i]]) for i in range(len(frames)))
    for i in (0, 1, 6, 9):
        ri = orc.project(frames[i], gb)
        gm = orc.ground_model(ri, tm, seed=3 + 100 + i)
        o = orc.compress_frame(frames[i], gb, tm, gm)
        assert _beq(outs[0]["ri"][i], o["range_image"]) and _beq(outs[0]["gm"][i], np.asarray(gm, np.float64)), i
        assert np.array_equal(outs[0]["seg"][i].reshape(-1), o["seg_idx"].reshape(-1).astype(np.uint8)), i
        assert np.array_equal(outs[0]["q"][i, :outs[0]["nnz"][i]], o["q"].astype(np.int16)), i


def test_random_orders_and_shapes(env):
    """Random batches through the probe and through the forced window kernel against the record kernels: frames of 1 .. 30 000 points, sorted by
    ring, by column, partly sorted, shuffled; duplicates; depth-0 points; four image shapes (whole image in the window, 2 and 4 windows per
    image, 128 rows)."""
    torch, ops, orc = env["torch"], env["ops"], env["orc"]
    rng = np.random.default_rng(606)
    shapes = [(16, 1800, 15.0, -15.0), (64, 2000, 2.0, -24.9), (64, 2048, 2.0, -24.9), (128, 2048, 15.0, -25.0), (40, 1024, 10.0, -20.0), (8, 16, 10.0, -10.0)]
    for draw in range(36):
        H, W, vmax, vmin = shapes[draw % len(shapes)]
        g = orc.LidarGeom(H=H, W=W, hfov_deg=360, vmax_deg=vmax, vmin_deg=vmin)
        geom = ops.make_geom(g.H, g.W, g.horizontal_FOV, g.vertical_max, g.vertical_min)
        B = int(rng.integers(1, 20))
        frames = []
        for _ in range(B):
            n = int(rng.integers(1, 30000))
            a = rng.normal(0, 15, (n, 3)).astype(np.float32)
            a[:, 2] = rng.normal(-1, 2.5, n)
            kind = rng.integers(0, 5)
            el = np.arctan2(a[:, 2], np.hypot(a[:, 0], a[:, 1]))
            az = np.arctan2(a[:, 1], a[:, 0])
            if kind == 0:
                a = a[np.lexsort((az, -np.round(el * 60)))]
            elif kind == 1:
                a = a[np.lexsort((el, np.round(az * 100)))]
            elif kind == 2:
                a = a[np.lexsort((az, -np.round(el * 60)))]
                k = n // 2
                a[k:] = a[k:][rng.permutation(n - k)]
            if n > 10 and rng.random() < 0.4:
                a[: n // 3] = a[n // 3: 2 * (n // 3)]
            if n > 10 and rng.random() < 0.25:
                a[rng.integers(0, n, 2)] = 0
            frames.append(a)
        ref, _ = _project(env, frames, geom, 0)
        for flags in (PROBE, FORCE):
            ri, acc = _project(env, frames, geom, flags, rows=bool(draw & 1))
            assert _beq(ri, ref), (draw, H, W, B, flags, acc)
        if draw % 6 == 0:
            i = int(rng.integers(0, B))
            assert _beq(ref[i], orc.project(frames[i], g)), (draw, i)

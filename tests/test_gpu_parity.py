"""-m gpu: the HIP path (through the C ABI) against the CPU oracle and the committed golden vectors.
Bit-exact for labels, indices and quantised integers; bit-exact for the fp32 range image / model /
prediction as well (the contract only asks for 1e-5 m there)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
MAN = json.load(open(os.path.join(HERE, "golden", "manifest.json")))


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
    assert np.array_equal(tm, orc.transform_map(g))
    return g, ops.make_geom(g.H, g.W, g.horizontal_FOV, g.vertical_max, g.vertical_min), tm


def _to(env, a):
    return env["torch"].from_numpy(np.ascontiguousarray(a)).to(env["dev"])


def _beq(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8))


@pytest.mark.parametrize("case", sorted(MAN["cases"]))
def test_golden_stage_by_stage(env, case):
    """Every stage entry point on the golden inputs vs the genuine-reference outputs."""
    torch, ops, orc = env["torch"], env["ops"], env["orc"]
    c = MAN["cases"][case]
    z = np.load(os.path.join(HERE, "golden", case + ".npz"))
    g, geom, tm = _geom(env, c["geom"])
    xyz, gm = z["xyz"], z["ground_model"]
    o = orc.compress_frame(xyz, g, tm, gm)          # oracle intermediates (pinned to the goldens on CPU)
    d_tm, d_gm = _to(env, tm), _to(env, gm.reshape(1, 4))
    offs = _to(env, np.array([0, xyz.shape[0]], np.int64))

    ri = ops.project(_to(env, xyz), offs, geom)
    assert _beq(ri[0].cpu().numpy(), o["range_image"])
    temp, info = ops.ground_mask(ri, d_tm, d_gm, 0.1)
    mask = (temp[0] > 0).cpu().numpy().reshape(g.H, g.W)
    assert np.array_equal(mask, o["mask"])
    inf = info[0].cpu().numpy()
    assert inf[0] == c["n_left"] and inf[2] == c["nnz"] and inf[1] == np.flatnonzero(o["mask"].reshape(-1))[0]

    cen_pix, centers = ops.fps_range(ri, d_tm, temp, info, 100)
    assert np.array_equal(cen_pix[0].cpu().numpy(), o["fps_pix"])
    assert _beq(centers[0].cpu().numpy(), o["centers"])
    # the reference's own op signature on the compacted candidate list
    pc_left = o["pc"][np.where(o["mask"])]
    idx = ops.fps_xyz(_to(env, pc_left[None]), 100)
    assert np.array_equal(idx[0].cpu().numpy(), o["fps_idx"])

    seg = ops.assign(ri, d_tm, d_gm, centers)
    assert np.array_equal(seg[0].cpu().numpy(), z["seg_idx"])
    model, counts = ops.point_model(ri, seg, d_gm, 100)
    nrow = int(z["seg_idx"].max()) + 1
    assert _beq(model[0, :nrow].cpu().numpy(), z["model_param"].astype(np.float32))
    assert np.array_equal(counts[0].cpu().numpy()[:nrow], np.bincount(z["seg_idx"].reshape(-1), minlength=nrow))
    q, nnz, pred = ops.predict_quantize(ri, d_tm, seg, model, 0.04, 100, want_pred=True)
    n = int(nnz[0])
    assert n == c["nnz"]
    assert _beq(pred[0].cpu().numpy().reshape(g.H, g.W, 1), o["pred"])
    assert np.array_equal(q[0, :n].cpu().numpy(), o["q"])
    assert np.array_equal(q[0, :n].cpu().numpy().astype(np.int16), z["q_uniform"])
    q16, nnz16, _ = ops.predict_quantize(ri, d_tm, seg, model, 0.04, 100, int16=True)
    assert np.array_equal(q16[0, :n].cpu().numpy(), z["q_uniform"])


def test_golden_fused_batch(env):
    """All four golden geometries... one geometry per batch: the fused entry on a 3-frame batch of the
    64x2048 golden frame (twice) plus a second synthetic frame; frame order must not matter."""
    torch, ops, orc, synth = env["torch"], env["ops"], env["orc"], env["synth"]
    z = np.load(os.path.join(HERE, "golden", "synth_64x2048.npz"))
    g, geom, tm = _geom(env, "Velodyne64E_2048")
    f2 = synth.make_frame(77, g.H, g.W).numpy()
    frames = [z["xyz"], f2, z["xyz"]]
    gms = np.stack([z["ground_model"], np.array([0.004, -0.01, -0.99994, -1.74]), z["ground_model"]])
    offs = np.zeros(4, np.int64)
    offs[1:] = np.cumsum([f.shape[0] for f in frames])
    buf = ops.BatchBuffers(3, geom, 100, env["dev"])
    ops.compress_batch(_to(env, np.concatenate(frames)), _to(env, offs), _to(env, tm), _to(env, gms), buf)
    torch.cuda.synchronize()
    for i in (0, 2):
        n = int(buf.nnz[i])
        assert np.array_equal(buf.seg[i].cpu().numpy(), z["seg_idx"])
        assert np.array_equal(buf.q16[i, :n].cpu().numpy(), z["q_uniform"])
        nrow = int(z["seg_idx"].max()) + 1
        assert _beq(buf.model[i, :nrow].cpu().numpy(), z["model_param"].astype(np.float32))
    o = orc.compress_frame(f2, g, tm, gms[1])
    n = int(buf.nnz[1])
    assert np.array_equal(buf.seg[1].cpu().numpy(), o["seg_idx"].astype(np.uint8))
    assert np.array_equal(buf.q16[1, :n].cpu().numpy(), o["q"].astype(np.int16))
    assert np.array_equal(buf.cen_pix[1].cpu().numpy(), o["fps_pix"])


def test_projection_edge_cases(env):
    """Pixel collisions (min depth), depth-0 points resetting a pixel in input order, denormal-small
    coordinates, points on the seams (azimuth wrap, elevation clamp), an empty frame and a 1-point frame."""
    torch, ops, orc = env["torch"], env["ops"], env["orc"]
    g, geom, tm = _geom(env, "Velodyne64E")
    rng = np.random.default_rng(21)
    a = rng.normal(0, 20, (60000, 3)).astype(np.float32)
    a[:, 2] = rng.normal(-1, 1.5, 60000)
    a[:3000] = a[3000:6000] * np.float32(1.00001)                 # collisions
    b = a.copy()
    for k in (100, 20000, 59999):                                 # depth-0 points at several positions
        b[k] = 0
    b[200] = [1e-30, -2e-30, 0]                                   # depth underflows to 0, azimuth != 0
    b[300:310] = [[5, -1e-9, 0.0]] * 10                           # azimuth just below 2*pi -> column wraps
    b[400] = [0, 0, 7]; b[401] = [0, 0, -7]                       # straight up / down: rows clamp
    c = np.zeros((0, 3), np.float32)
    d = np.array([[10.0, 2.0, -1.0]], np.float32)
    frames = [a, b, c, d, b[::-1].copy()]
    offs = np.zeros(len(frames) + 1, np.int64)
    offs[1:] = np.cumsum([f.shape[0] for f in frames])
    for atomic in (False, True):                                  # LDS-band path and device-atomic path
        ri = ops.project(_to(env, np.concatenate(frames)), _to(env, offs), geom, atomic_path=atomic).cpu().numpy()
        for i, f in enumerate(frames):
            assert _beq(ri[i], orc.project(f, g)), (atomic, i)
        assert not _beq(ri[1], ri[4])                             # input order matters with depth-0 points
    # non-finite points are skipped (documented deviation: the reference is undefined there)
    e = a[:1000].copy()
    e[5] = [np.nan, 1, 1]; e[6] = [np.inf, 1, 1]; e[7] = [1e30, 1e30, 0]
    keep = np.ones(1000, bool); keep[[5, 6, 7]] = False
    for atomic in (False, True):
        ri_e = ops.project(_to(env, e), _to(env, np.array([0, 1000], np.int64)), geom, atomic_path=atomic).cpu().numpy()
        assert _beq(ri_e[0], orc.project(e[keep], g))
    # a geometry whose image is not a whole number of LDS bands, real data with many pixel collisions
    z = np.load(os.path.join(HERE, "golden", "example_64E.npz"))
    g32, geom32, _ = _geom(env, "Velodyne32E")
    for atomic in (False, True):
        r = ops.project(_to(env, z["xyz"]), _to(env, np.array([0, z["xyz"].shape[0]], np.int64)), geom32,
                        atomic_path=atomic).cpu().numpy()
        assert _beq(r[0], orc.project(z["xyz"], g32))


def test_projection_rows_as_stored(env):
    """point_stride_bytes = 16: the sweep as a KITTI .bin stores it -- float32 rows (x, y, z, intensity), which the reference reads with
    np.fromfile(...).reshape(-1, 4) and slices [:, :3] on the host (dataset/dataset.py:48-50,62) -- goes to the kernels unsliced.  Same
    range image as the packed xyz, on the binned and on the device-atomic path, with depth-0 points (input-order fix-up), the
    exact-sequence queue, ragged / empty frames and a garbage 4th column (NaN, inf) that must never be read as a coordinate."""
    torch, ops, orc = env["torch"], env["ops"], env["orc"]
    g, geom, tm = _geom(env, "Velodyne64E")
    rng = np.random.default_rng(2104)
    a = rng.normal(0, 20, (70001, 3)).astype(np.float32)
    a[:, 2] = rng.normal(-1, 1.5, a.shape[0])
    b = a[:5003].copy()
    for k in (17, 2500, 5002):
        b[k] = 0                                                   # depth-0 points: the frame takes the fix-up path
    b[30:40] = [[5, -1e-9, 0.0]] * 10                             # column wrap: exact sequence
    frames = [a, b, np.zeros((0, 3), np.float32), a[:1].copy(), b[::-1].copy(), a[1000:3049].copy()]
    offs = np.zeros(len(frames) + 1, np.int64)
    offs[1:] = np.cumsum([f.shape[0] for f in frames])
    xyz = np.concatenate(frames)
    rows = np.empty((xyz.shape[0], 4), np.float32)
    rows[:, :3] = xyz
    rows[:, 3] = rng.choice(np.array([0.0, 0.37, np.nan, np.inf, -1e30], np.float32), xyz.shape[0])
    for atomic in (False, True):
        ri3 = ops.project(_to(env, xyz), _to(env, offs), geom, atomic_path=atomic).cpu().numpy()
        ri4 = ops.project(_to(env, rows), _to(env, offs), geom, atomic_path=atomic).cpu().numpy()
        assert _beq(ri3, ri4), atomic
        for i, f in enumerate(frames):
            assert _beq(ri4[i], orc.project(f, g)), (atomic, i)
    # the fused entry: every output of the batch
    gms = _to(env, np.tile(np.array([0.01, -0.02, -0.9997, -1.72]), (len(frames), 1)))
    outs = []
    for pts in (xyz, rows):
        buf = ops.BatchBuffers(len(frames), geom, 100, env["dev"])
        ops.compress_batch(_to(env, pts), _to(env, offs), _to(env, tm), gms.clone(), buf)
        torch.cuda.synchronize()
        outs.append([t.cpu().numpy().copy() for t in (buf.ri, buf.seg, buf.cen_pix, buf.nnz, buf.q16, buf.model)])
    for u, v in zip(*outs):
        if u.dtype == np.float32:
            assert _beq(u, v)
        elif u.dtype == np.int16:
            n = outs[0][3]
            assert all(np.array_equal(u[i, :n[i]], v[i, :n[i]]) for i in range(len(frames)))
        else:
            assert np.array_equal(u, v)
    # an unsupported stride and a misaligned row pointer are refused, not mis-read
    from rpcc_amd import _lib
    sc = torch.empty(_lib.lib().rpcc_project_scratch_bytes(8, 1, geom.H * geom.W), dtype=torch.uint8, device=env["dev"])
    ri = torch.empty((1, geom.H, geom.W), dtype=torch.float32, device=env["dev"])
    r16 = _to(env, rows[:9])
    o1 = _to(env, np.array([0, 8], np.int64))
    assert _lib.lib().rpcc_project_strided(ops.ptr(r16), 20, ops.ptr(o1), 8, 1, geom, ops.ptr(ri), ops.ptr(sc), sc.numel(), ops.stream()) == -1
    mis = r16.view(-1)[1:33]                                       # 4-byte aligned only
    assert _lib.lib().rpcc_project_strided(ops.ptr(mis), 16, ops.ptr(o1), 8, 1, geom, ops.ptr(ri), ops.ptr(sc), sc.numel(), ops.stream()) == -1


def test_projection_record_bins(env):
    """The pixel kernel bins its records by (frame, image band) in chunks that never cross a frame end: ragged batches of tiny and
    empty frames, a frame larger than one round of the band kernel's queue, scan-ordered points (one band per chunk), images of
    eight bands and of more bands than the binned path handles (device-atomic path), every point on the exact path."""
    torch, ops, orc, synth = env["torch"], env["ops"], env["orc"], env["synth"]
    rng = np.random.default_rng(77)

    def check(frames, g, geom, tag):
        offs = np.zeros(len(frames) + 1, np.int64)
        offs[1:] = np.cumsum([f.shape[0] for f in frames])
        xyz = np.concatenate(frames) if offs[-1] else np.zeros((0, 3), np.float32)
        ri = ops.project(_to(env, xyz) if offs[-1] else torch.zeros((0, 3), dtype=torch.float32, device=env["dev"]), _to(env, offs), geom).cpu().numpy()
        for i, f in enumerate(frames):
            assert _beq(ri[i], orc.project(f, g)), (tag, i)

    def cloud(n, spread=20.0):
        a = rng.normal(0, spread, (n, 3)).astype(np.float32)
        a[:, 2] = rng.normal(-1, 1.5, n)
        return a

    g, geom, _ = _geom(env, "Velodyne64E")
    # 1. sizes around the chunk size (2048) and its multiples, empty frames first, in the middle and last
    sizes = [0, 0, 1, 2047, 2048, 2049, 0, 4095, 4096, 4097, 63, 64, 65, 1, 0, 6000, 3, 0]
    check([cloud(n) for n in sizes], g, geom, "ragged")
    # 1b. a batch without any point
    check([np.zeros((0, 3), np.float32)] * 5, g, geom, "no points")
    # 2. many tiny frames (several per chunk of points)
    check([cloud(int(n)) for n in rng.integers(0, 300, 200)], g, geom, "tiny")
    # 3. a frame of more than 256 chunks (two rounds of the band kernel's queue) next to a small one
    check([cloud(2048 * 300 + 77), cloud(500)], g, geom, "large")
    # 4. scan order: consecutive points share an image row (all of a chunk's records fall into one band)
    gb = orc.LidarGeom(H=64, W=2048, hfov_deg=360, vmax_deg=2.0, vmin_deg=-24.9)
    geomb = ops.make_geom(gb.H, gb.W, gb.horizontal_FOV, gb.vertical_max, gb.vertical_min)
    f = synth.make_frame(3, 64, 2048).numpy()
    row = np.round((f[:, 2] / np.linalg.norm(f, axis=1)) * 200).astype(np.int64)      # (coarse elevation bins, then azimuth)
    order = np.lexsort((np.arctan2(f[:, 1], f[:, 0]), -row))
    check([f[order], f, f[order][::-1].copy()], gb, geomb, "scan order")
    # 5. eight bands (128 x 2048) and sixteen (128 x 4096: beyond the binned path)
    for W in (2048, 4096):
        g8 = orc.LidarGeom(H=128, W=W, hfov_deg=360, vmax_deg=15.0, vmin_deg=-25.0)
        geom8 = ops.make_geom(g8.H, g8.W, g8.horizontal_FOV, g8.vertical_max, g8.vertical_min)
        fr = [synth.make_frame(40 + i, 128, W, vmax_deg=15.0, vmin_deg=-25.0).numpy()[: 150000 + 1000 * i] for i in range(3)]
        check(fr + [cloud(5000, 8.0)], g8, geom8, "bands %d" % W)
    # 6. every point uncertain for the fast pixel test (on pixel boundaries of a coarse image): all records take the exact path
    gc = orc.LidarGeom(H=16, W=1800, hfov_deg=360, vmax_deg=15.0, vmin_deg=-15.0)
    geomc = ops.make_geom(gc.H, gc.W, gc.horizontal_FOV, gc.vertical_max, gc.vertical_min)
    az = (np.arange(20000) % 1800 + 0.5) * (2 * np.pi / 1800)          # exactly between two columns
    el = np.deg2rad(rng.uniform(-15, 15, 20000))
    r = rng.uniform(2, 60, 20000)
    pts = np.stack([r * np.cos(el) * np.cos(az), r * np.cos(el) * np.sin(az), r * np.sin(el)], 1).astype(np.float32)
    check([pts, pts[:100], pts[::-1].copy()], gc, geomc, "exact path")


def test_projection_paths_agree_on_random_batches(env):
    """The binned LDS-band path against the device-atomic path (both held to the oracle elsewhere) on random batches: 1 .. 48 frames
    of 0 .. 9000 points (clusters of tiny and empty frames included), four image shapes from 1 to 8 record bins, points in random
    or sorted order, duplicated points, depth-0 points sprinkled in."""
    torch, ops, orc = env["torch"], env["ops"], env["orc"]
    rng = np.random.default_rng(2024)
    shapes = [(16, 1800, 15.0, -15.0), (64, 2000, 2.0, -24.9), (64, 2048, 2.0, -24.9), (128, 2048, 15.0, -25.0), (7, 333, 10.0, -20.0)]
    for draw in range(60):
        H, W, vmax, vmin = shapes[draw % len(shapes)]
        g = orc.LidarGeom(H=H, W=W, hfov_deg=360, vmax_deg=vmax, vmin_deg=vmin)
        geom = ops.make_geom(g.H, g.W, g.horizontal_FOV, g.vertical_max, g.vertical_min)
        B = int(rng.integers(1, 49))
        frames = []
        for _ in range(B):
            kind = rng.integers(0, 5)
            n = 0 if kind == 0 else int(rng.integers(1, 40)) if kind == 1 else int(rng.integers(40, 9000))
            a = rng.normal(0, 15, (n, 3)).astype(np.float32)
            a[:, 2] = rng.normal(-1, 2.0, n)
            if n > 10 and rng.random() < 0.5:
                a[: n // 3] = a[n // 3: 2 * (n // 3)]                      # duplicates: pixel collisions with equal depths
            if n > 10 and rng.random() < 0.3:
                a[rng.integers(0, n, 3)] = 0                              # depth-0 points: the exact input-order path
            if n > 10 and rng.random() < 0.5:
                a = a[np.lexsort((np.arctan2(a[:, 1], a[:, 0]), -np.round(a[:, 2] * 4)))]
            frames.append(a)
        offs = np.zeros(B + 1, np.int64)
        offs[1:] = np.cumsum([f.shape[0] for f in frames])
        xyz = np.concatenate(frames) if offs[-1] else np.zeros((0, 3), np.float32)
        xt = _to(env, xyz) if offs[-1] else torch.zeros((0, 3), dtype=torch.float32, device=env["dev"])
        a_fast = ops.project(xt, _to(env, offs), geom)
        a_atom = ops.project(xt, _to(env, offs), geom, atomic_path=True)
        assert torch.equal(a_fast.view(torch.int32), a_atom.view(torch.int32)), (draw, H, W, B)
        if draw % 10 == 0:   # and the oracle now and then
            i = int(rng.integers(0, B))
            assert _beq(a_fast[i].cpu().numpy(), orc.project(frames[i], g)), (draw, i)


def test_fps_xyz_operator(env):
    """The reference FPS operator signature (B,N,3)->(B,M): ragged N, duplicates (exact ties), N<M."""
    torch, ops, orc = env["torch"], env["ops"], env["orc"]
    rng = np.random.default_rng(5)
    for (B, N, M) in [(3, 5000, 100), (2, 1023, 64), (1, 70, 100), (2, 4097, 17), (1, 1, 1), (1, 90000, 100), (2, 30000, 700),
                      (1, 2000, 2000)]:
        pts = rng.normal(0, 10, (B, N, 3)).astype(np.float32)
        pts[:, N // 2:] = pts[:, : N - N // 2]                    # exact duplicates -> distance ties
        pts[0, :min(N, 40)] = 0.0
        idx = ops.fps_xyz(_to(env, pts), M).cpu().numpy()
        for b in range(B):
            assert np.array_equal(idx[b], orc.fps(pts[b], M)), (B, N, M, b)


def test_fps_xyz_takes_the_kernel_the_point_order_suits(env):
    """rpcc_fps_xyz probes every list: consecutive points that are neighbours in space (the reference's row-major candidate list, a sweep in its
    stored order) run in the tile-pruned kernel, lists without locality in the one-pass-per-centre kernel -- per list, inside one call.  Both paths
    (and both load forms of each: N a multiple of four or not) against the brute-force entry and the oracle, incl. the final temp."""
    torch, ops, orc, synth = env["torch"], env["ops"], env["orc"], env["synth"]
    g, geom, tm = _geom(env, "Velodyne64E_2048")
    rng = np.random.default_rng(41)
    z = np.load(os.path.join(HERE, "golden", "example_64E.npz"))
    ri = orc.project(synth.make_frame(31, g.H, g.W).numpy(), g)
    rowmajor = orc.backproject(ri, tm).reshape(-1, 3)[ri.reshape(-1) != 0].astype(np.float32)   # pixels in row-major order: compact tiles
    stored = np.ascontiguousarray(z["xyz"], np.float32)                                          # the reference's example sweep as stored
    for base, name in ((rowmajor, "row-major"), (stored, "stored order")):
        for cut in (0, 3):                                   # N % 4 == 0 -> 16-byte loads; else the scalar forms
            n = (base.shape[0] // 4) * 4 - cut
            coherent = base[:n]
            shuffled = coherent[rng.permutation(n)]
            pts = np.stack([coherent, shuffled, coherent[::-1].copy()])
            marks = ops.fps_xyz_probe(_to(env, pts)).cpu().numpy()
            assert list(marks) == [0, -1, 0], (name, cut, marks)          # one call, both kernels
            out = {}
            for mode in (True, False):
                temp = torch.full((3, n), 1e10, dtype=torch.float32, device=env["dev"])
                idx = ops.fps_xyz(_to(env, pts), 100, temp=temp, bruteforce=mode)
                out[mode] = (idx.cpu().numpy(), temp.cpu().numpy())
            assert _beq(out[True][0], out[False][0]) and _beq(out[True][1], out[False][1]), (name, cut)
            for b in range(3):
                assert np.array_equal(out[False][0][b], orc.fps(pts[b], 100)), (name, cut, b)
    # small and degenerate lists: any decision is a correct one
    for pts in (np.zeros((2, 300, 3), np.float32), rng.normal(0, 1, (1, 5, 3)).astype(np.float32),
                np.full((1, 1024, 3), np.nan, np.float32)):
        m = min(pts.shape[1], 20)
        a = ops.fps_xyz(_to(env, pts), m).cpu().numpy()
        b = ops.fps_xyz(_to(env, pts), m, bruteforce=True).cpu().numpy()
        assert np.array_equal(a, b)


def test_point_model_sequential_fallback(env):
    """Ranges outside the fixed-point window [2^-5, 2^8) take the exact sequential fp64 path."""
    torch, ops, orc = env["torch"], env["ops"], env["orc"]
    rng = np.random.default_rng(9)
    H, W = 16, 1800
    seg = np.repeat(rng.integers(0, 102, (H, W // 8)), 8, axis=1).astype(np.uint8)
    seg[seg == 60] = 61                                            # an empty label -> NaN row
    ri = rng.uniform(0.5, 80, (H, W)).astype(np.float32)
    ri2 = ri.copy()
    ri2[3, 5:50] = rng.uniform(1e-4, 0.02, 45)                     # < 2^-5
    ri2[9, 100:140] = rng.uniform(300, 5000, 40)                   # >= 2^8
    gm = np.array([[0.01, -0.02, -0.999, -1.7]] * 2)
    for arr in (ri, ri2):
        arr[seg == 1] = 0
    model, counts = ops.point_model(_to(env, np.stack([ri, ri2])), _to(env, np.stack([seg, seg])), _to(env, gm), 100)
    for i, arr in enumerate((ri, ri2)):
        exp = orc.point_modeling(arr, seg.astype(np.int32))
        got = model[i].cpu().numpy()
        assert _beq(got[2:exp.shape[0], 3], exp[2:])
        assert got.view(np.uint32)[60, 3] == 0xFFC00000
        assert _beq(got[0], gm[i].astype(np.float32)) and not got[1].any()


def test_predict_quantize_with_plane_rows(env):
    """intra_predict's plane branch and the a+b+c==0 special case, plus int16 wrap-around."""
    torch, ops, orc = env["torch"], env["ops"], env["orc"]
    g, geom, tm = _geom(env, "Velodyne32E")
    rng = np.random.default_rng(10)
    seg = np.repeat(rng.integers(0, 102, (g.H, g.W // 10)), 10, axis=1).astype(np.uint8)
    ri = rng.uniform(0.5, 80, (g.H, g.W)).astype(np.float32)
    ri[seg == 1] = 0
    ri[4, 7] = 3000.0                                              # |q| > 32767 -> int16 wraps
    mp = np.zeros((102, 4))
    mp[:, 3] = rng.uniform(1, 60, 102)
    mp[0] = [0.0072, -0.054, -0.998, -1.76]
    for k in range(2, 102, 3):
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        mp[k] = [n[0], n[1], n[2], -rng.uniform(2, 30)]
    mp[5] = [0.5, -0.5, 0.0, 7.0]
    mp[1] = 0
    q, nnz, pred = ops.predict_quantize(_to(env, ri[None]), _to(env, tm), _to(env, seg[None]),
                                        _to(env, mp.astype(np.float32)[None]), 0.04, 100, want_pred=True)
    pr = orc.intra_predict(seg.astype(np.int32), mp, tm)
    assert _beq(pred[0].cpu().numpy().reshape(g.H, g.W, 1), pr)
    qo = orc.uniform_quantize(seg.astype(np.int32), ri.reshape(g.H, g.W, 1) - pr, 0.04)
    n = int(nnz[0])
    assert n == qo.shape[0] and np.array_equal(q[0, :n].cpu().numpy(), qo)
    q16, _, _ = ops.predict_quantize(_to(env, ri[None]), _to(env, tm), _to(env, seg[None]),
                                     _to(env, mp.astype(np.float32)[None]), 0.04, 100, int16=True)
    assert np.array_equal(q16[0, :n].cpu().numpy(), qo.astype(np.int16))


def test_full_size_batch_properties(env):
    """BASELINE config[1] shape: a 64-frame batch of 64x2048 sweeps through the fused entry.
    Size-independent properties on every frame + oracle equality on a sample."""
    torch, ops, orc, synth = env["torch"], env["ops"], env["orc"], env["synth"]
    g, geom, tm = _geom(env, "Velodyne64E_2048")
    B = 64
    xyz, offs = synth.make_batch(range(1000, 1000 + B), g.H, g.W, device=env["dev"])
    rng = np.random.default_rng(3)
    gms = np.tile(np.array([0.0, 0.0, -1.0, -1.73]), (B, 1)) + rng.normal(0, 0.004, (B, 4))
    buf = ops.BatchBuffers(B, geom, 100, env["dev"])
    d_tm = _to(env, tm)
    ops.compress_batch(xyz, offs, d_tm, _to(env, gms), buf)
    torch.cuda.synchronize()
    seg = buf.seg.cpu().numpy(); ri = buf.ri.cpu().numpy(); nnz = buf.nnz.cpu().numpy()
    counts = buf.counts.cpu().numpy(); info = buf.info.cpu().numpy()
    model = buf.model.cpu().numpy(); q16 = buf.q16.cpu().numpy()
    P = g.H * g.W
    for b in range(B):
        assert nnz[b] == (ri[b] != 0).sum() == info[b, 2]
        assert np.array_equal(seg[b] == 1, ri[b] == 0)
        assert np.array_equal(counts[b], np.bincount(seg[b].reshape(-1), minlength=102))
        assert len(set(buf.cen_pix[b].cpu().numpy().tolist())) == 100
        # decode: dequantise in label order and check the reconstruction bound (README.md:101-106)
        pred = orc.intra_predict(seg[b].astype(np.int32), model[b].astype(np.float64), tm)[..., 0]
        rec = np.zeros((g.H, g.W), np.float32)
        order = np.argsort(seg[b].reshape(-1), kind="stable")
        order = order[seg[b].reshape(-1)[order] != 1]
        rec.reshape(-1)[order] = q16[b, :nnz[b]].astype(np.float32) * np.float32(0.04)
        err = np.abs((pred + rec) - ri[b])[ri[b] != 0]
        assert err.max() <= 0.02 + 1e-5, (b, err.max())
    xyz_c, offs_c = xyz.cpu().numpy(), offs.cpu().numpy()
    for b in (0, 31, 63):
        o = orc.compress_frame(xyz_c[offs_c[b]:offs_c[b + 1]], g, tm, gms[b])
        assert np.array_equal(seg[b], o["seg_idx"].astype(np.uint8))
        assert np.array_equal(q16[b, :nnz[b]], o["q"].astype(np.int16))
    # idempotence: a second run over the same inputs gives identical bytes
    q_first = q16.copy()
    ops.compress_batch(xyz, offs, d_tm, _to(env, gms), buf)
    torch.cuda.synchronize()
    assert np.array_equal(buf.q16.cpu().numpy()[:, :nnz.min()], q_first[:, :nnz.min()])


def test_assign_sqrt_ties_and_duplicates(env):
    """argmax(-abs(distance)) keeps the LOWEST index among clusters whose fp32 radius is equal: that
    includes squared distances that differ by an ulp but round to the same sqrtf, exact duplicates of a
    centre, and ties between the ground term and a cluster (ground wins)."""
    torch, ops, orc = env["torch"], env["ops"], env["orc"]
    g, geom, tm = _geom(env, "VelodyneVLP16")
    rng = np.random.default_rng(31)
    ri = rng.uniform(2, 60, (g.H, g.W)).astype(np.float32)
    ri[rng.random(ri.shape) < 0.1] = 0
    pc = orc.backproject(ri, tm).reshape(-1, 3)
    M = 100
    cen = pc[rng.choice(pc.shape[0], M, replace=False)].copy()
    # centres mirrored about chosen pixels, the LOWER index nudged by a few ulps: near-ties under sqrtf
    n_sqrt_ties = 0
    pix = rng.choice(pc.shape[0], 40, replace=False)
    for j, p in enumerate(pix):
        if j >= 40:
            break
        v = rng.normal(0, 0.4, 3).astype(np.float32)
        hi, lo = 99 - j, 2 * j if 2 * j < 50 else j
        cen[hi] = pc[p] + v
        cb = pc[p] - v
        cb[rng.integers(0, 3)] *= np.float32(1 + rng.integers(-3, 4) * 2.0 ** -23)
        cen[lo] = cb
    cen[60] = cen[10]                                              # exact duplicate centre
    plane = np.array([0.01, -0.02, -0.9997, -1.72])
    exp = orc.assign(ri, pc.reshape(g.H, g.W, 3), tm, plane, cen)
    d = pc[:, None, :] - cen[None, :, :]
    d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    rad = np.sqrt(d2)
    kmin = d2.argmin(1)
    n_sqrt_ties = int(((rad == rad[np.arange(len(kmin)), kmin][:, None]) & (d2 != d2[np.arange(len(kmin)), kmin][:, None])).any(1).sum())
    assert n_sqrt_ties > 0, "test inputs must contain sqrt-level ties"
    got = ops.assign(_to(env, ri[None]), _to(env, tm), _to(env, plane[None]), _to(env, cen[None]))[0].cpu().numpy()
    assert np.array_equal(got, exp.astype(np.uint8))
    assert np.array_equal(exp, orc.np_assign(ri.reshape(g.H, g.W, 1), pc.reshape(g.H, g.W, 3), tm, plane, cen))


def test_ground_mask_threshold_boundary(env):
    """a5: depth_dif > threshold with the threshold set exactly ON, one ulp below and one ulp above the fp64 quotient
    of chosen pixels (the kernel screens the division and must divide for exactly these), plus degenerate planes and
    thresholds; with and without the fused FPS table."""
    torch, ops, orc = env["torch"], env["ops"], env["orc"]
    g, geom, tm = _geom(env, "VelodyneVLP16")
    rng = np.random.default_rng(5)
    ri = rng.uniform(2, 40, (g.H, g.W)).astype(np.float32)
    ri[rng.random(ri.shape) < 0.1] = 0
    pc = orc.backproject(ri, tm)
    cases = []
    for plane in (np.array([0.01, -0.02, -0.9997, -1.72]), np.array([3.0, -1.0, -250.0, -431.0]), np.array([1e-9, 2e-9, -3e-8, -5e-8])):
        dd = orc.vertical_residual(pc, plane).reshape(-1)
        for p in rng.choice(dd.size, 6, replace=False):
            for thr in (dd[p], np.nextafter(dd[p], 0), np.nextafter(dd[p], np.inf)):
                cases.append((plane, float(thr)))
    cases += [(np.zeros(4), 0.1), (np.array([0.0, 0.0, -1.0, -1.7]), 0.0), (np.array([0.0, 0.0, -1.0, -1.7]), -0.5),
              (np.array([0.0, 0.0, -1e200, -1e200]), 0.1), (np.array([np.nan, 0.0, -1.0, -1.7]), 0.1),
              (np.array([0.0, 0.0, -1.0, -1.7]), 1e-300)]
    n_on = 0
    for plane, thr in cases:
        with np.errstate(all="ignore"):
            exp = orc.vertical_residual(pc, plane).reshape(-1) > thr
        n_on += int((orc.vertical_residual(pc, plane).reshape(-1) == thr).sum())
        for tab in (False, True):
            out = ops.ground_mask(_to(env, ri[None]), _to(env, tm), _to(env, plane[None]), thr, fps_table=tab)
            temp, info = out[0], out[1]
            assert np.array_equal(temp[0].cpu().numpy() >= 0, exp), (plane, thr, tab)
            assert int(info[0, 0]) == int(exp.sum())
    assert n_on >= 18


def test_assign_ground_screen_band(env):
    """a7: the fp32 screen of the ground term (DESIGN.md "assign") must hand every pixel whose cluster radius lies
    inside its error band to the fp64 sequence: centres placed at a distance equal to the pixel's ground residual
    up to relative offsets of 0 .. 1e-5, planes that are not normalised, nearly parallel to rays, through the
    origin, or not finite."""
    torch, ops, orc = env["torch"], env["ops"], env["orc"]
    g, geom, tm = _geom(env, "Velodyne32E")
    rng = np.random.default_rng(77)
    ri = rng.uniform(3, 50, (g.H, g.W)).astype(np.float32)
    ri[rng.random(ri.shape) < 0.05] = 0
    pc = orc.backproject(ri, tm).reshape(-1, 3)
    tmf = tm.reshape(-1, 3).astype(np.float64)
    M = 100
    planes = [np.array([0.01, -0.02, -0.9997, -1.72]), np.array([10.0, -20.0, -999.7, -1720.0]),
              np.array([1e-3, 2e-3, -0.5e-3, -1e-3]), np.array([0.0, 0.3, -0.05, -0.4]), np.array([0.02, 0.01, -1.0, 0.0]),
              np.array([np.nan, 0.0, -1.0, -1.7]), np.array([0.0, 0.0, -1.0, np.inf]), np.array([0.0, 0.0, 0.0, -1.7])]
    n_band = 0
    for plane in planes:
        cen = pc[rng.choice(pc.shape[0], M, replace=False)].copy()
        with np.errstate(all="ignore"):
            den = (tmf[:, 0] * plane[0] + tmf[:, 1] * plane[1]) + tmf[:, 2] * plane[2]
            ag = np.abs(ri.reshape(-1).astype(np.float64) - (-plane[3] / den))
        ok = np.flatnonzero(np.isfinite(ag) & (ag > 0.05) & (ag < 30) & (ri.reshape(-1) != 0))
        if ok.size >= M:
            pix = rng.choice(ok, M, replace=False)
            for j, p in enumerate(pix):
                u = rng.normal(0, 1, 3)
                u /= np.linalg.norm(u)
                delta = [0.0, 1e-8, -1e-8, 1e-7, -1e-7, 1e-6, -1e-6, 1e-5, -1e-5, 3e-8][j % 10]
                cen[j] = (pc[p].astype(np.float64) + u * ag[p] * (1 + delta)).astype(np.float32)
            d = pc[pix].astype(np.float32) - cen
            rad = np.sqrt(((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]).astype(np.float32)).astype(np.float64)
            n_band += int((np.abs(rad - ag[pix]) < 1e-5 * ag[pix]).sum())
        exp = orc.assign(ri, pc.reshape(g.H, g.W, 3), tm, plane, cen)
        got = ops.assign(_to(env, ri[None]), _to(env, tm), _to(env, plane[None]), _to(env, cen[None]))[0].cpu().numpy()
        assert np.array_equal(got, exp.astype(np.uint8)), plane
    assert n_band > 100, "test inputs must contain pixels inside the screen's error band"


@pytest.mark.parametrize("fma,cuda_tie", [(0, True), (1, False), (1, True), (2, False), (2, True)])
def test_fps_cuda_binary_modes_equal_the_oracle(env, fma, cuda_tie):
    """a6: the CUDA binary's contraction of sampling_gpu.cu:64 and its reduction tree's winner among equal values as selectable
    modes (RPCC_FPS_FMA1 / _FMA2 / _TIE_CUDA): the kernel equals oracle.fps_modes (itself checked against a thread-by-thread
    restatement of the CUDA kernel) -- on point lists made of ties, ragged sizes incl. powers of two, on a range image whose
    constant range makes whole rings of pixels equidistant, on a synthetic sweep through the stage entries and through the fused
    entry, and via the environment variables the front-ends read."""
    torch, ops, orc, synth = env["torch"], env["ops"], env["orc"], env["synth"]
    rng = np.random.default_rng(100 + 10 * fma + int(cuda_tie))
    # (1) operator seam: lattices (ties everywhere), sizes around the block-size rule of opt_n_threads
    for n, m in ((7, 5), (64, 20), (1000, 40), (1024, 33), (1025, 33), (2048, 64), (3000, 100), (20000, 60)):
        pts = rng.integers(-4, 5, (2, n, 3)).astype(np.float32)
        got = ops.fps_xyz(_to(env, pts), m, fma=fma, cuda_tie=cuda_tie).cpu().numpy()
        for b in range(2):
            assert np.array_equal(got[b], orc.fps_modes(pts[b], m, fma, cuda_tie)), (n, m, b)
    pts = rng.normal(0, 20, (1, 30000, 3)).astype(np.float32)
    assert np.array_equal(ops.fps_xyz(_to(env, pts), 100, fma=fma, cuda_tie=cuda_tie).cpu().numpy()[0], orc.fps_modes(pts[0], 100, fma, cuda_tie))
    # (2) range images: constant range (rings of exactly equidistant pixels, empty pixels = one class of equal points) and a sweep
    g, geom, tm = _geom(env, "VelodyneVLP16")
    gd = orc.GEOMS["VelodyneVLP16"]
    ri_c = np.full((g.H, g.W), 12.0, np.float32)
    ri_c[rng.random(ri_c.shape) < 0.15] = 0
    xyz = synth.make_frame(4242, g.H, g.W, vmax_deg=gd["vmax_deg"], vmin_deg=gd["vmin_deg"]).numpy()
    ri_s = orc.project(xyz, g)
    plane = np.array([0.01, -0.02, -0.9997, -1.72])
    M = 40
    for ri in (ri_c, ri_s):
        cfg = dict(orc.DEFAULT_CFG, cluster_num=M, fps_fma=fma, fps_cuda_tie=cuda_tie)
        want = orc.segment(ri, tm, plane, cfg)
        d_ri, d_tm, d_pl = _to(env, ri[None]), _to(env, tm), _to(env, plane[None])
        temp, info = ops.ground_mask(d_ri, d_tm, d_pl, 0.1, fps_table=False)
        pix, cen = ops.fps_range(d_ri, d_tm, temp, info, M, fma=fma, cuda_tie=cuda_tie)
        assert np.array_equal(pix[0].cpu().numpy(), want["fps_pix"])
        assert _beq(cen[0].cpu().numpy(), want["centers"])
        if cuda_tie:   # the tie rule is exercised: the default rule picks other pixels on the constant-range image
            base = orc.segment(ri, tm, plane, dict(cfg, fps_cuda_tie=False))
            if ri is ri_c:
                assert not np.array_equal(base["fps_pix"], want["fps_pix"])
    # (3) fused entry with the flags, and the environment variables of the front-ends
    frames = [xyz, synth.make_frame(4243, g.H, g.W, vmax_deg=gd["vmax_deg"], vmin_deg=gd["vmin_deg"]).numpy()]
    offs = np.zeros(3, np.int64)
    offs[1:] = np.cumsum([f.shape[0] for f in frames])
    ground = np.stack([plane, np.array([-0.01, 0.005, -0.9999, -1.75])])
    buf = ops.BatchBuffers(2, geom, 100, env["dev"])
    old = {k: os.environ.get(k) for k in ("RPCC_FPS_FMA", "RPCC_FPS_TIE_CUDA")}
    try:
        for via_env in (False, True):
            if via_env:
                os.environ["RPCC_FPS_FMA"], os.environ["RPCC_FPS_TIE_CUDA"] = str(fma), "1" if cuda_tie else "0"
            ops.compress_batch(_to(env, np.concatenate(frames)), _to(env, offs), _to(env, tm), _to(env, ground), buf,
                               fps_fma=None if via_env else fma, fps_cuda_tie=None if via_env else cuda_tie)
            for i, f in enumerate(frames):
                o = orc.compress_frame(f, g, tm, ground[i], dict(orc.DEFAULT_CFG, fps_fma=fma, fps_cuda_tie=cuda_tie))
                assert np.array_equal(buf.cen_pix[i].cpu().numpy(), o["fps_pix"]), (via_env, i)
                assert np.array_equal(buf.seg[i].cpu().numpy(), o["seg_idx"].astype(np.uint8))
                n = int(buf.nnz[i])
                assert n == o["q"].shape[0] and np.array_equal(buf.q16[i, :n].cpu().numpy(), o["q"].astype(np.int16))
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)


def test_fps_tiled_equals_bruteforce(env):
    """The tile-pruned FPS kernels are exact: same indices, same centres AND the same final temp array
    (bit for bit) as the brute-force kernels, on range images and on explicit point lists."""
    torch, ops, orc, synth = env["torch"], env["ops"], env["orc"], env["synth"]
    for name, ids in (("Velodyne64E_2048", (5, 6)), ("Velodyne64E", (7,)), ("Velodyne32E", (8,)), ("VelodyneVLP16", (9,))):
        g, geom, tm = _geom(env, name)
        gd = orc.GEOMS[name]
        frames = [synth.make_frame(i, g.H, g.W, vmax_deg=gd["vmax_deg"], vmin_deg=gd["vmin_deg"]).numpy() for i in ids]
        offs = np.zeros(len(frames) + 1, np.int64)
        offs[1:] = np.cumsum([f.shape[0] for f in frames])
        ri = ops.project(_to(env, np.concatenate(frames)), _to(env, offs), geom)
        gms = _to(env, np.tile(np.array([0.002, -0.004, -0.99999, -1.73]), (len(frames), 1)))
        # one more frame whose first 64 pixels lie ON the ground plane: none of them is a candidate, so the
        # "first FPS pass inside ground_mask" shortcut must decline (info[b][3] == 0) and fall back
        if name == "Velodyne64E_2048":
            pl = gms[0].cpu().numpy()
            r_plane = (-pl[3] / (tm[0, :64].astype(np.float64) @ pl[:3])).astype(np.float32)
            assert (r_plane > 0).all()
            ri = torch.cat([ri, ri[:1].clone()], 0)
            ri[-1, 0, :64] = _to(env, r_plane)
            gms = torch.cat([gms, gms[:1]], 0)
        res = {}
        for mode in ("brute", "tiled", "tiled+table"):
            if mode == "tiled+table":
                temp, info, tab = ops.ground_mask(ri, _to(env, tm), gms, 0.1, fps_table=True)
                flags = info[:, 3].cpu().numpy()
                assert flags[0] == 1
                if name == "Velodyne64E_2048":
                    assert flags[-1] == 0
            else:
                temp, info = ops.ground_mask(ri, _to(env, tm), gms, 0.1)
                tab = None
            # info[:, 4]: the first empty pixel that is a candidate (the representative of the FPS origin class)
            ri_h, t_h = ri.cpu().numpy().reshape(len(gms), -1), temp.cpu().numpy()
            for i in range(len(gms)):
                e = np.flatnonzero((ri_h[i] == 0) & (t_h[i] >= 0))
                assert int(info[i, 4]) == (int(e[0]) if e.size else g.H * g.W), (name, mode, i)
            cen_pix, centers = ops.fps_range(ri, _to(env, tm), temp, info, 100, fps_table=tab, bruteforce=(mode == "brute"))
            res[mode] = (cen_pix.cpu().numpy(), centers.cpu().numpy(), temp.cpu().numpy())
        for mode in ("tiled", "tiled+table"):
            for a, b in zip(res["brute"], res[mode]):
                assert _beq(a, b), (name, mode)
        res[False] = res["tiled+table"]
        for i, f in enumerate(frames):
            o = orc.compress_frame(f, g, tm, gms[i].cpu().numpy())
            assert np.array_equal(res[False][0][i], o["fps_pix"])
    rng = np.random.default_rng(77)
    for (B, N, M) in [(2, 30000, 100), (1, 777, 50), (1, 64, 10), (1, 200000, 100), (1, 300000, 20)]:
        pts = rng.normal(0, 10, (B, N, 3)).astype(np.float32)
        pts[:, N // 3: 2 * (N // 3)] = pts[:, : N // 3]
        out = {}
        for mode in (True, False):
            temp = torch.full((B, N), 1e10, dtype=torch.float32, device=env["dev"])
            idx = ops.fps_xyz(_to(env, pts), M, temp=temp, bruteforce=mode)
            out[mode] = (idx.cpu().numpy(), temp.cpu().numpy())
        assert _beq(out[True][0], out[False][0]) and _beq(out[True][1], out[False][1]), (B, N, M)
        if N <= 30000:
            assert np.array_equal(out[False][0][0], orc.fps(pts[0], M))


def test_ground_ransac_matches_specification(env):
    """a4: the seeded ground RANSAC kernel equals the sequential form of its specification (oracle)
    bit for bit -- >5000 candidates (systematic subsample), 800..5000 (all candidates) and <800
    (every pixel) -- and the plane is a sensible ground plane."""
    torch, ops, orc, synth = env["torch"], env["ops"], env["orc"], env["synth"]
    g, geom, tm = _geom(env, "Velodyne64E_2048")
    f0 = synth.make_frame(11, g.H, g.W).numpy()
    f1 = f0[f0[:, 2] > -1.45]                                                    # no ground at all
    low = np.flatnonzero(f0[:, 2] < -1.5)
    f2 = np.concatenate([f0[f0[:, 2] >= -1.5], f0[low[:3000]]])                  # ~3000 candidates
    z = np.load(os.path.join(HERE, "golden", "example_64E.npz"))
    for frames, name in (([f0, f1, f2], "Velodyne64E_2048"), ([z["xyz"]], "Velodyne64E")):
        g, geom, tm = _geom(env, name)
        offs = np.zeros(len(frames) + 1, np.int64)
        offs[1:] = np.cumsum([f.shape[0] for f in frames])
        ri = ops.project(_to(env, np.concatenate(frames)), _to(env, offs), geom)
        ground, inl = ops.ground_ransac(ri, _to(env, tm), seed=40)
        ground, inl = ground.cpu().numpy(), inl.cpu().numpy()
        for i, f in enumerate(frames):
            rio = orc.project(f, g)
            cand = orc.ground_candidates(rio, tm)
            pl, n = orc.ransac_plane(cand, 10, 100, 0.1, 40 + i)
            assert _beq(ground[i], pl), (name, i, ground[i], pl)
            assert inl[i] == n
        assert abs(abs(ground[0][2]) - 1) < 0.01 and abs(abs(ground[0][3]) - 1.73) < 0.1
    # fused entry with the ground fit inside == stage-by-stage with that model injected
    g, geom, tm = _geom(env, "Velodyne64E_2048")
    fz = f0.copy()
    fz[100] = 0                                     # depth-0 point: exact re-projection, the fit counts for itself
    frames = [f0, f2, fz, f2[:500].copy()]          # the band kernel hands the candidate counts to the fit (zcnt)
    offs = np.zeros(len(frames) + 1, np.int64)
    offs[1:] = np.cumsum([f.shape[0] for f in frames])
    buf = ops.BatchBuffers(len(frames), geom, 100, env["dev"])
    gfit = torch.zeros((len(frames), 4), dtype=torch.float64, device=env["dev"])
    ops.compress_batch(_to(env, np.concatenate(frames)), _to(env, offs), _to(env, tm), gfit, buf, ground_seed=7)
    torch.cuda.synchronize()
    for i, f in enumerate(frames):
        gm = orc.ground_model(orc.project(f, g), tm, seed=7 + i)
        assert _beq(gfit[i].cpu().numpy(), gm)
        o = orc.compress_frame(f, g, tm, gm)
        n = int(buf.nnz[i])
        assert np.array_equal(buf.seg[i].cpu().numpy(), o["seg_idx"].astype(np.uint8))
        assert np.array_equal(buf.q16[i, :n].cpu().numpy(), o["q"].astype(np.int16))


def _np_dequantize(q, seg, steps, salience=None):
    """dequantize_residual, utils/compress_utils.py:114-132 (numpy, as the reference writes it)."""
    residual = np.zeros_like(seg, dtype=np.float32)
    start = 0
    for m in range(int(seg.max()) + 1):
        idx = np.where(seg == m)
        if m == 1:
            continue
        cur = steps if salience is None else steps[salience[m]]
        residual[idx] = q[start:start + idx[0].shape[0]] * cur
        start += idx[0].shape[0]
    assert start == q.shape[0]
    return np.expand_dims(residual, -1)


@pytest.mark.parametrize("case", sorted(MAN["cases"]))
def test_contour_codec_and_decoder(env, case):
    """f1: contour bits / index sequence == the reference payload; f3: recover_map, dequantise, predict,
    reconstruct == the reference decoder arithmetic; reconstruction error <= accuracy."""
    import hashlib
    torch, ops, orc = env["torch"], env["ops"], env["orc"]
    c = MAN["cases"][case]
    z = np.load(os.path.join(HERE, "golden", case + ".npz"))
    g, geom, tm = _geom(env, c["geom"])
    seg_np = z["seg_idx"]
    seg = _to(env, seg_np[None])
    bits, seq, nseq = ops.contour_encode(seg)
    n = int(nseq[0])
    cm, sq = orc.extract_contour(seg_np.astype(np.int32))
    assert n == sq.shape[0]
    assert np.array_equal(bits[0].cpu().numpy(), np.packbits(cm.astype(bool), axis=None))
    assert np.array_equal(seq[0, :n].cpu().numpy(), sq.astype(np.uint16))
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    assert sha(bits[0].cpu().numpy()) == c["sha"]["contour_map"]
    assert sha(seq[0, :n].cpu().numpy()) == c["sha"]["idx_sequence"]
    # decoder
    seg_rec = ops.contour_decode(bits, seq, g.H, g.W)
    assert np.array_equal(seg_rec[0].cpu().numpy(), seg_np)
    mp32 = z["model_param"].astype(np.float32)
    model = np.zeros((1, 102, 4), np.float32)
    model[0, :mp32.shape[0]] = mp32
    q = z["q_uniform"]
    q_pad = np.zeros((1, g.H * g.W), np.int16)
    q_pad[0, :q.shape[0]] = q
    rec, pc = ops.decode(seg_rec, _to(env, q_pad), _to(env, model), _to(env, tm), 0.04, want_points=True)
    pred = orc.intra_predict(seg_np.astype(np.int32), mp32, tm)
    exp = pred + _np_dequantize(q, seg_np, 0.04)
    assert _beq(rec[0].cpu().numpy().reshape(g.H, g.W, 1), exp)
    assert _beq(pc[0].cpu().numpy(), exp * tm)
    ri = orc.project(z["xyz"], g)
    err = np.abs(rec[0].cpu().numpy() - ri)[ri != 0]
    assert err.max() <= 0.02 + 1e-5
    # non-uniform steps
    sal = z["salience"]
    salp = np.zeros((1, 102), np.uint8)
    salp[0, :sal.shape[0]] = sal
    steps = np.array([0.04] * 4) + np.array([0, 0.02, 0.04, 0.06])
    qn = z["q_nonuniform"]
    qn_pad = np.zeros((1, g.H * g.W), np.int16)
    qn_pad[0, :qn.shape[0]] = qn
    rec_n, _ = ops.decode(seg_rec, _to(env, qn_pad), _to(env, model), _to(env, tm), steps, salience=_to(env, salp))
    exp_n = pred + _np_dequantize(qn, seg_np, steps, sal)
    assert _beq(rec_n[0].cpu().numpy().reshape(g.H, g.W, 1), exp_n)
    # a3 entry
    assert _beq(ops.backproject(_to(env, ri[None]), _to(env, tm))[0].cpu().numpy(), orc.backproject(ri, tm))


def test_contour_codec_ragged_rows(env):
    """Widths that are not multiples of 8 / 64 / 1024 and the reference's own known-answer vector."""
    torch, ops, orc = env["torch"], env["ops"], env["orc"]
    k = MAN["contour_kat"]
    rng = np.random.default_rng(12)
    maps = [np.array(k["idx_map"], np.uint8)]
    for (h, w) in [(3, 5), (7, 13), (16, 1800), (5, 1031), (64, 2000)]:
        maps.append(np.repeat(rng.integers(0, 102, (h, (w + 6) // 7)), 7, axis=1)[:, :w].astype(np.uint8))
    for mp in maps:
        seg = _to(env, mp[None])
        bits, seq, nseq = ops.contour_encode(seg)
        cm, sq = orc.extract_contour(mp.astype(np.int32))
        n = int(nseq[0])
        assert n == sq.shape[0] and np.array_equal(seq[0, :n].cpu().numpy(), sq.astype(np.uint16)), mp.shape
        assert np.array_equal(bits[0].cpu().numpy(), np.packbits(cm.astype(bool), axis=None)), mp.shape
        assert np.array_equal(ops.contour_decode(bits, seq, mp.shape[0], mp.shape[1])[0].cpu().numpy(), mp), mp.shape
    assert orc.extract_contour(np.array(k["idx_map"]))[1].tolist() == k["idx_sequence"]


@pytest.mark.parametrize("case", sorted(MAN["cases"]))
def test_nonuniform_framework(env, case):
    """a12 key points, a13 salience levels + per-label quantisation == the reference C++ run with a
    zero-initialised key point map (golden q_nonuniform / salience / key_point_map)."""
    torch, ops, orc = env["torch"], env["ops"], env["orc"]
    c = MAN["cases"][case]
    z = np.load(os.path.join(HERE, "golden", case + ".npz"))
    g, geom, tm = _geom(env, c["geom"])
    ri_np = orc.project(z["xyz"], g)
    seg_np = z["seg_idx"]
    ri, seg = _to(env, ri_np[None]), _to(env, seg_np[None])
    feat, kp = ops.extract_features(ri, seg)
    feat_o, kp_o = orc.extract_features_with_segment(ri_np, seg_np.astype(np.int32))
    assert np.array_equal(kp[0].cpu().numpy(), z["key_point_map"])
    assert np.array_equal(kp[0].cpu().numpy(), kp_o.astype(np.uint8))
    assert _beq(feat[0].cpu().numpy(), feat_o)
    lacc = (np.array([0.04] * 4) + np.array([0, 0.02, 0.04, 0.06])).astype(np.float32)
    sal, label_acc = ops.salience(seg, kp, [30, 10, 3, 0], lacc, 2, 100)
    nrow = z["salience"].shape[0]
    assert np.array_equal(sal[0, :nrow].cpu().numpy(), z["salience"])
    mp = np.zeros((1, 102, 4), np.float32)
    mp[0, :nrow] = z["model_param"].astype(np.float32)
    q, nnz, _ = ops.predict_quantize(ri, _to(env, tm), seg, _to(env, mp), 0.04, 100, int16=True, label_acc=label_acc)
    n = int(nnz[0])
    assert n == z["q_nonuniform"].shape[0]
    assert np.array_equal(q[0, :n].cpu().numpy(), z["q_nonuniform"])
    # the quantiser's own seam: residual handed in by the caller (utils/compress_utils.py:57)
    pred = ops.intra_predict(seg, _to(env, mp), _to(env, tm))
    assert _beq(pred[0].cpu().numpy().reshape(g.H, g.W, 1), orc.intra_predict(seg_np.astype(np.int32), mp[0], tm))
    res = (ri - pred).reshape(1, -1).contiguous()
    q2, _, _ = ops.predict_quantize(ri, _to(env, tm), seg, _to(env, mp), 0.04, 100, residual=res)
    assert np.array_equal(q2[0, :n].cpu().numpy().astype(np.int16), z["q_uniform"])


def test_features_edge_rows(env):
    """Rows with too few valid pixels are skipped; ties in curvature; gaps trigger the occlusion gate;
    odd widths."""
    torch, ops, orc = env["torch"], env["ops"], env["orc"]
    rng = np.random.default_rng(15)
    for (H, W) in [(8, 301), (16, 1800), (4, 64), (3, 2250)]:
        seg = np.repeat(rng.integers(0, 20, (H, (W + 4) // 5)), 5, axis=1)[:, :W].astype(np.uint8)
        ri = (15 + 4 * np.sin(np.arange(W) / 9.0)[None, :] + rng.normal(0, 0.03, (H, W))).astype(np.float32)
        ri[:, ::53] += 2.5
        ri[0, :] = 20.0                       # constant row: every curvature is exactly 0
        seg[1, :] = 0                         # no valid pixel at all
        seg[2, 10:] = 1                       # fewer than segments + 2*fr + 1 valid pixels
        ri[seg == 1] = 0
        feat, kp = ops.extract_features(_to(env, ri[None]), _to(env, seg[None]))
        f_o, k_o = orc.extract_features_with_segment(ri, seg.astype(np.int32))
        assert np.array_equal(kp[0].cpu().numpy(), k_o.astype(np.uint8)), (H, W)
        assert _beq(feat[0].cpu().numpy(), f_o), (H, W)
        assert k_o.max() >= 1


FEATURE_PARAM_DRAWS = [(3, 8, 4, 8, 6)] + [tuple(int(v) for v in r) for r in np.stack([
    np.random.default_rng(77).integers(1, 6, 14), np.random.default_rng(78).integers(2, 13, 14),
    np.random.default_rng(79).integers(0, 7, 14), np.random.default_rng(80).integers(0, 13, 14),
    np.random.default_rng(81).integers(0, 11, 14)], 1)]     # the draws tests/test_oracle_vs_ref.py pins against the reference's C++


@pytest.mark.parametrize("params", FEATURE_PARAM_DRAWS)
def test_features_and_salience_parameter_sweep(env, params):
    """a12 / a13 away from the YAML defaults: feature_region, segments, sharp / less_sharp / flat counts (zeros and
    less_sharp < sharp included), then salience levels with drawn key-point thresholds, level count and ground level."""
    torch, ops, orc = env["torch"], env["ops"], env["orc"]
    fr, segments, sharp, less, flat = params
    rng = np.random.default_rng(1000 + 7 * fr + segments)
    for (h, w) in [(48, 1500), (7, 333)]:
        seg = np.repeat(rng.integers(0, 30, (h, (w + 5) // 6)), 6, axis=1)[:, :w].astype(np.int32)
        ri = (20 + 5 * np.sin(np.arange(w) / 23.0)[None, :] + rng.normal(0, 0.04, (h, w))).astype(np.float32)
        ri[:, ::61] += 2.0
        ri[3, :] = 17.0
        seg[5, 40:] = 1
        ri[seg == 1] = 0
        f_o, k_o = orc.extract_features_with_segment(ri, seg, fr, segments, sharp, less, flat)
        seg_d = _to(env, seg.astype(np.uint8)[None])
        feat, kp = ops.extract_features(_to(env, ri[None]), seg_d, fr, segments, sharp, less, flat)
        assert np.array_equal(kp[0].cpu().numpy(), k_o.astype(np.uint8)), (params, h, w)
        assert _beq(feat[0].cpu().numpy(), f_o), (params, h, w)
        levels = int(rng.integers(1, 7))
        lk = np.sort(rng.integers(0, 40, levels))[::-1].astype(np.int32)
        lk[-1] = 0                                                       # the last level accepts everything
        la = (0.04 + np.sort(rng.uniform(0, 0.1, levels))).astype(np.float32)
        gl = int(rng.integers(0, levels))
        res = rng.normal(0, 0.3, (h, w, 1)).astype(np.float32)
        q_o, s_o = orc.nonuniform_quantize(seg, res, k_o, lk, la, gl)
        sal, label_acc = ops.salience(seg_d, kp, lk, la, gl, 100)
        assert np.array_equal(sal[0, : s_o.shape[0]].cpu().numpy(), s_o.astype(np.uint8)), (params, h, w)
        q, nnz, _ = ops.predict_quantize(_to(env, ri[None]), None, seg_d, None, 0.04, 100, label_acc=label_acc,
                                         residual=_to(env, res.reshape(1, -1)))
        n = int(nnz[0])
        assert n == q_o.shape[0] and np.array_equal(q[0, :n].cpu().numpy(), q_o), (params, h, w)


@pytest.mark.parametrize("case,angle", [("example_64E", 75), ("synth_64x2048", 75), ("synth_vlp16", 75), ("synth_32E", 40),
                                        ("example_64E", 20)])
def test_plane_model(env, case, angle):
    """a9: per-label plane model == cluster_modeling('plane') of the reference with the build's seeded
    RANSAC in Open3D's place: plane rows, angle rejection -> numpy fp32 mean, < 30 pixel labels, empty
    labels.  Then the planes are used for prediction + quantisation like tools/compress.py does."""
    torch, ops, orc = env["torch"], env["ops"], env["orc"]
    c = MAN["cases"][case]
    z = np.load(os.path.join(HERE, "golden", case + ".npz"))
    g, geom, tm = _geom(env, c["geom"])
    ri_np = orc.project(z["xyz"], g)
    seg_np = z["seg_idx"].copy()
    seg_np[seg_np == 40] = 41                                    # an empty label
    small = np.flatnonzero(seg_np.reshape(-1) == 50)
    seg_np.reshape(-1)[small[12:]] = 51                          # a label with < 30 pixels
    pc = orc.backproject(ri_np, tm)
    ri2 = np.stack([ri_np, ri_np])
    seg2 = np.stack([seg_np, seg_np])
    gm = np.stack([z["ground_model"], z["ground_model"]])
    model, counts = ops.plane_model(_to(env, ri2), _to(env, tm), _to(env, seg2), 100, angle_threshold=angle, seed=5,
                                    ground=_to(env, gm), want_counts=True)
    model = model.cpu().numpy()
    nrow = int(seg_np.max()) + 1
    n_plane = n_mean = 0
    for b in range(2):                                           # frame index enters the per-label seed
        exp = orc.cluster_modeling_plane(pc, ri_np, seg_np.astype(np.int64), tm, angle_deg=angle, seed=5, frame=b)
        exp32 = exp.astype(np.float32)
        assert _beq(model[b, 1:nrow], exp32), (case, b)
        assert _beq(model[b, 0], gm[b].astype(np.float32))
        n_plane += int((exp[:, :3] != 0).any(1).sum())
        n_mean += int(((exp[:, :3] == 0).all(1) & (exp[:, 3] != 0)).sum())
    assert model.view(np.uint32)[0, 40, 3] == 0xFFC00000
    assert n_mean > 0 and (n_plane > 0 or angle < 30)
    assert np.array_equal(counts[0].cpu().numpy()[:nrow], np.bincount(seg_np.reshape(-1), minlength=nrow))
    # downstream: prediction with plane rows + quantisation == oracle with the same rows
    mp = np.concatenate((gm[:1], orc.cluster_modeling_plane(pc, ri_np, seg_np.astype(np.int64), tm, angle, 5, 0)), 0)
    q, nnz, pred = ops.predict_quantize(_to(env, ri_np[None]), _to(env, tm), _to(env, seg_np[None]), _to(env, model[:1]),
                                        0.04, 100, want_pred=True)
    pr = orc.intra_predict(seg_np.astype(np.int32), mp, tm)
    assert _beq(pred[0].cpu().numpy().reshape(g.H, g.W, 1), pr)
    qo = orc.uniform_quantize(seg_np.astype(np.int32), ri_np.reshape(g.H, g.W, 1) - pr, 0.04)
    assert np.array_equal(q[0, : int(nnz[0])].cpu().numpy(), qo)


@pytest.mark.parametrize("gname", ["Velodyne64E_2048", "Velodyne64E", "Velodyne32E", "VelodyneVLP16"])
def test_projection_fast_path_never_disagrees(env, gname):
    """a2: the screened fast pixel computation (DESIGN.md "Projection") must agree with the exact fdlibm/IEEE sequence
    on every point it claims to be certain about -- random, lidar-like, boundary-adversarial and special-value inputs,
    ~1e8 points per geometry, compared on the device (rpcc_project_fastpath_check).  The exact sequence itself is
    pinned against the oracle / golden vectors by the other projection tests."""
    torch, ops = env["torch"], env["ops"]
    g, geom, _ = _geom(env, gname)
    dev = env["dev"]
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    n = 1 << 25

    def rnd(*shape):
        return torch.rand(*shape, generator=gen, device=dev, dtype=torch.float64)

    def dirs(az, el, r):
        return torch.stack((r * torch.cos(el) * torch.cos(az), r * torch.cos(el) * torch.sin(az), r * torch.sin(el)), 1).float()

    vfov = g.vertical_max - g.vertical_min
    sets = {}
    sets["cube"] = ((rnd(n, 3) - 0.5) * 160).float()
    # lidar-like: every direction of the sweep, jittered inside the pixel
    az = rnd(n) * g.horizontal_FOV
    el = g.vertical_min + rnd(n) * vfov
    sets["sweep"] = dirs(az, el, 1 + rnd(n) * 100)
    # adversarial: directions within +-2e-5 rad of a column / row rounding boundary
    kc = torch.randint(0, g.W, (n,), generator=gen, device=dev).double()
    kr = torch.randint(0, g.H, (n,), generator=gen, device=dev).double()
    eps = (rnd(n) - 0.5) * 4e-5
    az_b = (kc + 0.5) * (g.horizontal_FOV / g.W) + eps
    el_b = g.vertical_min + (kr + 0.5) * (vfov / (g.H - 1)) + (rnd(n) - 0.5) * 4e-5
    sets["col_boundary"] = dirs(az_b, el, 1 + rnd(n) * 100)
    sets["row_boundary"] = dirs(az, el_b, 1 + rnd(n) * 100)
    sets["above_below"] = dirs(az, (rnd(n) - 0.5) * 3.1, 1 + rnd(n) * 50)           # outside the vertical FOV: clamped rows
    # magnitudes from 1e-30 to 1e30, signs, exact zeros, x == 1, axis-aligned points
    mag = torch.pow(10.0, (rnd(n, 3) - 0.5) * 60)
    sgn = torch.where(rnd(n, 3) < 0.5, -1.0, 1.0)
    sp = (mag * sgn).float()
    sp[::7, 0] = 0.0
    sp[1::11, 1] = 0.0
    sp[2::13, 2] = -0.0
    sp[3::17, 0] = 1.0
    sp[4::19] = 0.0
    sp[5::23, 1] = float("inf")
    sp[6::29, 2] = float("nan")
    sets["special"] = sp
    tot_sure = 0
    for name, xyz in sets.items():
        sure, bad, slow, dcol, drow = ops.project_fastpath_check(xyz.contiguous(), geom)
        assert sure + slow == xyz.shape[0]
        assert bad == 0, (name, sure, bad, slow)
        # the margins are 4x the analytic worst case (pix_fast_cfg: PIX_MARGIN): the observed discrepancy must stay well below the
        # budget itself -- below a QUARTER of it, i.e. the margin is at least 16 x what any of these 8e8 points shows
        bcol = 2.0e-6 * g.W / g.horizontal_FOV + 6.0e-7 * g.W
        brow = 2.1e-6 * (g.H - 1) / vfov + 6.0e-7 * (g.H + abs(g.vertical_min) * (g.H - 1) / vfov)
        assert dcol < bcol / 4 and drow < brow / 4, (name, dcol, bcol, drow, brow)
        print("fastpath %-13s %-16s sure %.4f  dcol %.2e (budget %.2e)  drow %.2e (budget %.2e)"
              % (name, gname, sure / xyz.shape[0], dcol, bcol, drow, brow))
        tot_sure += sure
        if name == "sweep":
            assert slow < 0.03 * n, (name, slow / n)     # the fast path must carry the ordinary points (1.8 % uncertain with directions spread evenly)
    assert tot_sure > 0


@pytest.mark.parametrize("H,W,M", [(5, 300, 7), (8, 512, 20), (33, 1000, 100), (16, 4000, 50), (128, 2048, 30), (128, 4096, 30),
                                   (7, 301, 9), (12, 1030, 40), (20, 2051, 60)])   # widths that are no multiple of four: quads at 4-byte alignment, row-end quads by element
def test_odd_geometries_fused(env, H, W, M):
    """Fused entry (ground fit inside) on image shapes that are no multiple of any tile size the kernels use
    (4x32 FPS / assign tiles, 1024-pixel scatter tiles, 32768-pixel projection bands, RANSAC chunks), with
    cluster counts from 7 to 100, and on images of 8 record bins (the binned projection's limit) and of 16 (the fused entry
    then projects with device atomics and a separate initialisation launch): every output equals the oracle's."""
    torch, ops, orc, synth = env["torch"], env["ops"], env["orc"], env["synth"]
    g = orc.LidarGeom(H, W, 360.0, 3.0, -25.0)
    tm = ops.transform_map(g.H, g.W, g.horizontal_FOV, g.vertical_max, g.vertical_min)
    assert np.array_equal(tm, orc.transform_map(g))
    geom = ops.make_geom(g.H, g.W, g.horizontal_FOV, g.vertical_max, g.vertical_min)
    frames = [synth.make_frame(300 + i, H, W, vmax_deg=3.0, vmin_deg=-25.0).numpy() for i in range(3)]
    frames.append(frames[0][:40].copy())                          # nearly empty frame: fewer candidates than clusters
    offs = np.zeros(len(frames) + 1, np.int64)
    offs[1:] = np.cumsum([f.shape[0] for f in frames])
    B = len(frames)
    buf = ops.BatchBuffers(B, geom, M, env["dev"])
    gfit = torch.zeros((B, 4), dtype=torch.float64, device=env["dev"])
    ops.compress_batch(_to(env, np.concatenate(frames)), _to(env, offs), _to(env, tm), gfit, buf, ground_seed=11)
    torch.cuda.synchronize()
    cfg = dict(orc.DEFAULT_CFG)
    cfg["cluster_num"] = M
    for i, f in enumerate(frames[:3]):
        gm = orc.ground_model(orc.project(f, g), tm, seed=11 + i)
        assert _beq(gfit[i].cpu().numpy(), gm), (H, W, i)
        o = orc.compress_frame(f, g, tm, gm, cfg)
        n = int(buf.nnz[i])
        assert _beq(buf.ri[i].cpu().numpy(), o["range_image"])
        assert np.array_equal(buf.cen_pix[i].cpu().numpy(), o["fps_pix"]), (H, W, i)
        assert np.array_equal(buf.seg[i].cpu().numpy(), o["seg_idx"].astype(np.uint8)), (H, W, i)
        assert n == o["q"].shape[0] and np.array_equal(buf.q16[i, :n].cpu().numpy(), o["q"].astype(np.int16))
    # the sparse frame: projection and ground fit still agree (FPS with fewer candidates than clusters is undefined
    # in the reference -- indices repeat -- so only the stages before it are compared)
    gm = orc.ground_model(orc.project(frames[3], g), tm, seed=11 + 3)
    assert _beq(gfit[3].cpu().numpy(), gm)
    assert _beq(buf.ri[3].cpu().numpy(), orc.project(frames[3], g))


@pytest.mark.parametrize("gname,nframes", [("VelodyneVLP16", 40), ("Velodyne32E", 16)])
def test_fused_batch_many_frames_vs_oracle(env, gname, nframes):
    """Breadth check of the exact screens (projection fast path, assign ground / tie screens, ground-mask division): every
    frame of a larger batch of different synthetic scenes, ground plane fitted inside the call, equals the oracle in
    range image, FPS pixels, labels and quantised integers."""
    torch, ops, orc, synth = env["torch"], env["ops"], env["orc"], env["synth"]
    g, geom, tm = _geom(env, gname)
    gd = orc.GEOMS[gname]
    frames = [synth.make_frame(1000 + 7 * i, g.H, g.W, vmax_deg=gd["vmax_deg"], vmin_deg=gd["vmin_deg"]).numpy() for i in range(nframes)]
    offs = np.zeros(nframes + 1, np.int64)
    offs[1:] = np.cumsum([f.shape[0] for f in frames])
    buf = ops.BatchBuffers(nframes, geom, 100, env["dev"])
    gfit = torch.zeros((nframes, 4), dtype=torch.float64, device=env["dev"])
    ops.compress_batch(_to(env, np.concatenate(frames)), _to(env, offs), _to(env, tm), gfit, buf, ground_seed=500)
    torch.cuda.synchronize()
    gf = gfit.cpu().numpy()
    ri, seg, cen, q16, nnz = (buf.ri.cpu().numpy(), buf.seg.cpu().numpy(), buf.cen_pix.cpu().numpy(), buf.q16.cpu().numpy(),
                              buf.nnz.cpu().numpy())
    for i, f in enumerate(frames):
        gm = orc.ground_model(orc.project(f, g), tm, seed=500 + i)
        assert _beq(gf[i], gm), (gname, i)
        o = orc.compress_frame(f, g, tm, gm)
        assert _beq(ri[i], o["range_image"]), (gname, i)
        assert np.array_equal(cen[i], o["fps_pix"]), (gname, i)
        assert np.array_equal(seg[i], o["seg_idx"].astype(np.uint8)), (gname, i)
        n = int(nnz[i])
        assert n == o["q"].shape[0] and np.array_equal(q16[i, :n], o["q"].astype(np.int16)), (gname, i)


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("RPCC_FUZZ_SEEDS", "24")))))
def test_fuzz_fused_vs_oracle(env, seed):
    """Randomised breadth: image shape, fields of view, cluster count, accuracy and ground threshold drawn per seed; the
    synthetic scene is rescaled / tilted and salted with duplicates, far points, points at the origin and on the optical
    axis.  Three frames per draw through the fused entry; every integer output and the range image equal the oracle's."""
    torch, ops, orc, synth = env["torch"], env["ops"], env["orc"], env["synth"]
    rng = np.random.default_rng(9000 + seed)
    H = int(rng.integers(4, 41))
    W = int(rng.integers(96, 1500))
    vmax = float(rng.uniform(1.0, 16.0))
    vmin = float(-rng.uniform(10.0, 31.0))
    hfov = float(rng.choice([360.0, 360.0, 180.0, 90.0]))
    M = int(rng.integers(3, 61))
    accuracy = float(rng.choice([0.01, 0.02, 0.05, 0.1]))
    thr = float(rng.choice([0.05, 0.1, 0.2]))
    g = orc.LidarGeom(H, W, hfov, vmax, vmin)
    tm = ops.transform_map(g.H, g.W, g.horizontal_FOV, g.vertical_max, g.vertical_min)
    assert np.array_equal(tm, orc.transform_map(g))
    geom = ops.make_geom(g.H, g.W, g.horizontal_FOV, g.vertical_max, g.vertical_min)
    frames = []
    for i in range(3):
        f = synth.make_frame(7000 + 10 * seed + i, H, W, vmax_deg=vmax, vmin_deg=vmin, hfov_deg=hfov).numpy()
        f = f * np.float32(rng.uniform(0.5, 1.5))                                   # nearer / farther scene
        a = np.float32(rng.uniform(-0.03, 0.03))                                    # small roll
        f = np.stack([f[:, 0], f[:, 1] * np.cos(a) - f[:, 2] * np.sin(a), f[:, 1] * np.sin(a) + f[:, 2] * np.cos(a)], 1).astype(np.float32)
        extra = [f[rng.integers(0, len(f), 50)],                                    # exact duplicates
                 f[rng.integers(0, len(f), 50)] * np.float32(1.0000001),            # near-duplicates (same pixel, other depth)
                 (rng.normal(size=(20, 3)) * 300).astype(np.float32),               # far points, any direction
                 np.zeros((3, 3), np.float32),                                      # the origin (depth 0)
                 np.array([[0, 0, 5], [0, 0, -5], [1e-30, 0, 1]], np.float32)]      # on / next to the vertical axis
        f = np.concatenate([f] + extra).astype(np.float32)
        frames.append(f[rng.permutation(len(f))])
    frames.insert(int(rng.integers(0, 4)), np.zeros((0, 3), np.float32))            # a frame without any point, anywhere in the batch
    offs = np.zeros(len(frames) + 1, np.int64)
    offs[1:] = np.cumsum([f.shape[0] for f in frames])
    B = len(frames)
    buf = ops.BatchBuffers(B, geom, M, env["dev"])
    gfit = torch.zeros((B, 4), dtype=torch.float64, device=env["dev"])
    ops.compress_batch(_to(env, np.concatenate(frames)), _to(env, offs), _to(env, tm), gfit, buf, ground_threshold=thr,
                       acc=2 * accuracy, ground_seed=40 + seed)
    torch.cuda.synchronize()
    cfg = dict(orc.DEFAULT_CFG, cluster_num=M, accuracy=accuracy, ground_threshold=thr)
    tag = (seed, H, W, M)
    compared = 0
    for i, f in enumerate(frames):
        if len(f) == 0:                                                            # empty sweep: empty image, every pixel label 1, no payload
            assert not buf.ri[i].any().item() and (buf.seg[i] == 1).all().item() and int(buf.nnz[i]) == 0, tag
            continue
        ri_o = orc.project(f, g)
        assert _beq(buf.ri[i].cpu().numpy(), ri_o), tag
        gm = orc.ground_model(ri_o, tm, seed=40 + seed + i)
        assert _beq(gfit[i].cpu().numpy(), gm), tag
        o = orc.compress_frame(f, g, tm, gm, cfg)
        if len(set(o["fps_pix"].tolist())) < M:
            continue                                                               # fewer candidates than clusters: undefined in the reference
        n = int(buf.nnz[i])
        assert np.array_equal(buf.cen_pix[i].cpu().numpy(), o["fps_pix"]), tag
        assert np.array_equal(buf.seg[i].cpu().numpy(), o["seg_idx"].astype(np.uint8)), tag
        assert _beq(buf.model[i, : o["model_param"].shape[0]].cpu().numpy(), o["model_param"].astype(np.float32)), tag
        assert n == o["q"].shape[0] and np.array_equal(buf.q16[i, :n].cpu().numpy(), o["q"].astype(np.int16)), tag
        compared += 1
    assert compared >= 1, tag


@pytest.mark.gpu
@pytest.mark.parametrize("B,P", [(1, 7), (3, 2048), (37, 4099), (5, 4096), (6, 8192), (256, 131072)])
def test_pack_payload(env, B, P):
    """f2: the frames' int16 runs back to back (prefix sums on the device), incl. empty frames, full frames, a capacity
    that cuts the stream, and -- at the bench size -- the stream of a real batch."""
    torch, ops = env["torch"], env["ops"]
    rng = np.random.default_rng(B * 1000 + P)
    q = rng.integers(-32768, 32768, (B, P), dtype=np.int64).astype(np.int16)
    nnz = rng.integers(0, P + 1, B).astype(np.int32)
    nnz[0] = P
    if B > 2:
        nnz[1], nnz[-1] = 0, P
    if B > 4:
        nnz[2], nnz[3] = P - 1, P                                          # odd prefix in front of a full frame
    want = np.concatenate([q[b, : nnz[b]] for b in range(B)])
    packed, total = ops.pack_payload(_to(env, q), _to(env, nnz))
    assert int(total.item()) == want.shape[0]
    assert np.array_equal(packed[: want.shape[0]].cpu().numpy(), want)
    assert not packed[want.shape[0]:].any().item()                      # nothing written past the stream
    cap = max(want.shape[0] // 2, 1)
    buf = torch.full((cap + 64,), 77, dtype=torch.int16, device=packed.device)
    ops.pack_payload(_to(env, q), _to(env, nnz), packed=buf, capacity=cap, total=total)
    assert int(total.item()) == want.shape[0]                           # total is the stream length, not what fitted
    assert np.array_equal(buf[:cap].cpu().numpy(), want[:cap]) and (buf[cap:] == 77).all().item()


@pytest.mark.parametrize("angle", [75, 0.001])
def test_plane_model_large_labels(env, angle):
    """a9 on labels of every size class of the kernel: merged labels of tens of thousands of pixels (workgroup path, several
    8192-element blocks of NumPy's pairwise mean), a few thousand (around the workgroup / wavefront switch) and the
    ordinary small ones; angle 0.001 rejects every plane, so every label goes through the fp32 mean."""
    torch, ops, orc = env["torch"], env["ops"], env["orc"]
    case = "synth_64x2048"
    c = MAN["cases"][case]
    z = np.load(os.path.join(HERE, "golden", case + ".npz"))
    g, geom, tm = _geom(env, c["geom"])
    ri_np = orc.project(z["xyz"], g)
    seg_np = z["seg_idx"].copy()
    seg_np[seg_np >= 60] = 60
    seg_np[(seg_np >= 10) & (seg_np < 20)] = 10
    seg_np[(seg_np >= 30) & (seg_np < 33)] = 30
    cnt = np.bincount(seg_np.reshape(-1), minlength=102)
    assert cnt[60] > 3 * 8192 and cnt[2:].min() == 0 and (cnt[2:] > 2048).sum() >= 2
    pc = orc.backproject(ri_np, tm)
    gm = z["ground_model"][None]
    # cluster_num 59 -> 61 model rows: the last workgroup of the frame (four labels each) is partly empty
    model = ops.plane_model(_to(env, ri_np[None]), _to(env, tm), _to(env, seg_np[None]), 59, angle_threshold=angle, seed=9,
                            ground=_to(env, gm)).cpu().numpy()
    exp = orc.cluster_modeling_plane(pc, ri_np, seg_np.astype(np.int64), tm, angle_deg=angle, seed=9, frame=0).astype(np.float32)
    nrow = int(seg_np.max()) + 1
    assert _beq(model[0, 1:nrow], exp)
    if angle < 1:
        assert not (exp[1:, :3] != 0).any()
    else:
        assert (exp[:, :3] != 0).any(1).sum() > 5


def _mixed_groups(env, specs, M=100):
    """specs: [(lidar name, frames, keyword overrides)] -> (list of compress_batch argument dicts on fresh buffers, frame lists)."""
    torch, ops, orc, synth = env["torch"], env["ops"], env["orc"], env["synth"]
    groups = []
    for k, (name, n, over) in enumerate(specs):
        g, geom, tm = _geom(env, name)
        gd = orc.GEOMS[name]
        frames = [synth.make_frame(5100 + 31 * k + i, g.H, g.W, vmax_deg=gd["vmax_deg"], vmin_deg=gd["vmin_deg"]).numpy() for i in range(n)]
        offs = np.zeros(n + 1, np.int64)
        offs[1:] = np.cumsum([f.shape[0] for f in frames])
        general = over.get("model_method", "point") != "point" or over.get("nonuniform") is not None
        a = dict(xyz=_to(env, np.concatenate(frames)), offsets=_to(env, offs), tm=_to(env, tm),
                 ground=torch.zeros((n, 4), dtype=torch.float64, device=env["dev"]), buf=ops.BatchBuffers(n, geom, M, env["dev"], general=general),
                 ground_seed=70 + k, frame_ids=_to(env, np.arange(900 + 10 * k, 900 + 10 * k + n, dtype=np.int64)))
        a.update(over)
        if a["ground_seed"] < 0:      # injected ground models
            a["ground"] = _to(env, np.tile(np.array([[0.01, -0.02, 1.0, 1.7]]), (n, 1)))
        groups.append(a)
    return groups


_MIXED_CASES = {
    "uniform_point": lambda ops: [("Velodyne64E", 3, {}), ("Velodyne32E", 2, {}), ("VelodyneVLP16", 4, {})],
    "nonuniform_plane": lambda ops: [(n, b, dict(model_method="plane", plane_seed=5, nonuniform=ops.nonuniform_cfg(0.04)))
                                     for n, b in (("VelodyneVLP16", 3), ("Velodyne64E", 2), ("Velodyne32E", 3))],
    # groups that differ in everything a group may differ in: injected ground, the brute-force FPS kernel, a CUDA-binary FPS mode, model, framework
    "every_group_different": lambda ops: [("Velodyne32E", 2, dict(ground_seed=-1)), ("VelodyneVLP16", 3, dict(fps_bruteforce=True, model_method="plane")),
                                          ("Velodyne64E", 2, dict(nonuniform=ops.nonuniform_cfg(0.04))), ("VelodyneVLP16", 2, dict(fps_fma=1))],
    "one_group": lambda ops: [("Velodyne32E", 3, dict(model_method="plane"))],
}


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(_MIXED_CASES))
def test_compress_batch_mixed_equals_the_groups_alone(env, case):
    """rpcc_compress_batch_mixed (variable H x W inside one call; the ground RANSAC, FPS and plane-fit launches shared by the groups):
    every output buffer of every group is bit for bit what rpcc_compress_batch gives for that group alone -- which the tests above
    hold to the oracle -- for the uniform / point and non-uniform / plane paths, for groups whose settings differ, and for one group."""
    torch, ops = env["torch"], env["ops"]
    specs = _MIXED_CASES[case](ops)
    alone = _mixed_groups(env, specs)
    for a in alone:
        ops.compress_batch(**a)
    mixed = _mixed_groups(env, specs)
    ops.compress_batch_mixed(mixed)
    torch.cuda.synchronize()
    for k, (a, m) in enumerate(zip(alone, mixed)):
        ba, bm = a["buf"], m["buf"]
        assert _beq(a["ground"].cpu().numpy(), m["ground"].cpu().numpy()), (case, k)
        for f in ("ri", "seg", "cen_pix", "centers", "model", "counts", "nnz"):
            assert _beq(getattr(ba, f).cpu().numpy(), getattr(bm, f).cpu().numpy()), (case, k, f)
        nz = ba.nnz.cpu().numpy()
        qa, qm = ba.q16.cpu().numpy(), bm.q16.cpu().numpy()
        assert all(np.array_equal(qa[i, :nz[i]], qm[i, :nz[i]]) for i in range(ba.B)), (case, k)
        if a.get("nonuniform") is not None:
            assert np.array_equal(ba.salience.cpu().numpy(), bm.salience.cpu().numpy()) and np.array_equal(ba.key_point_map.cpu().numpy(), bm.key_point_map.cpu().numpy())


@pytest.mark.gpu
def test_compress_batch_mixed_argument_errors(env):
    """More groups than RPCC_MAX_GROUPS, differing cluster counts and a missing output buffer are refused before anything is launched."""
    ops = env["ops"]
    g5 = _mixed_groups(env, [("VelodyneVLP16", 1, {})] * 5)
    with pytest.raises(AssertionError):
        ops.compress_batch_mixed(g5)
    two = _mixed_groups(env, [("VelodyneVLP16", 1, {})]) + _mixed_groups(env, [("VelodyneVLP16", 1, {})], M=50)
    with pytest.raises(AssertionError):
        ops.compress_batch_mixed(two)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(8)))
def test_fuzz_mixed_call_vs_groups_alone(env, seed):
    """Randomised mixed batches: one to four geometry groups with random image shapes (odd widths, few rows), frame counts, cluster
    count, thresholds and per-group settings (ground fitted or injected, point / plane model, uniform / non-uniform framework, brute-force
    FPS): rpcc_compress_batch_mixed == rpcc_compress_batch per group, every output buffer, bit for bit."""
    torch, ops, orc, synth = env["torch"], env["ops"], env["orc"], env["synth"]
    rng = np.random.default_rng(4200 + seed)
    G = int(rng.integers(1, 5))
    M = int(rng.integers(5, 101))
    acc = float(rng.choice([0.02, 0.04, 0.1]))
    thr = float(rng.choice([0.05, 0.1, 0.2]))

    def build():
        r = np.random.default_rng(77 + seed)     # the same draws for both runs
        groups = []
        for k in range(G):
            H, W = int(r.integers(4, 41)), int(r.integers(96, 1500))
            vmax, vmin = float(r.uniform(1.0, 16.0)), float(-r.uniform(10.0, 31.0))
            n = int(r.integers(1, 7))
            g = orc.LidarGeom(H, W, 360.0, vmax, vmin)
            tm = ops.transform_map(g.H, g.W, g.horizontal_FOV, g.vertical_max, g.vertical_min)
            geom = ops.make_geom(g.H, g.W, g.horizontal_FOV, g.vertical_max, g.vertical_min)
            frames = [synth.make_frame(8800 + 100 * seed + 10 * k + i, H, W, vmax_deg=vmax, vmin_deg=vmin).numpy() for i in range(n)]
            offs = np.zeros(n + 1, np.int64)
            offs[1:] = np.cumsum([f.shape[0] for f in frames])
            plane, nonuni, inject, brute = (bool(r.integers(0, 2)) for _ in range(4))
            a = dict(xyz=_to(env, np.concatenate(frames)), offsets=_to(env, offs), tm=_to(env, tm),
                     ground=torch.zeros((n, 4), dtype=torch.float64, device=env["dev"]),
                     buf=ops.BatchBuffers(n, geom, M, env["dev"], general=plane or nonuni), ground_seed=int(r.integers(0, 1000)),
                     frame_ids=_to(env, r.integers(0, 1 << 20, n).astype(np.int64)), model_method="plane" if plane else "point",
                     plane_seed=int(r.integers(0, 1000)), nonuniform=ops.nonuniform_cfg(acc) if nonuni else None, fps_bruteforce=brute and H * W % 4 == 0)
            if inject:
                a["ground_seed"] = -1
                a["ground"] = _to(env, np.tile(np.array([[0.02, 0.01, 1.0, 1.7]]), (n, 1)))
            groups.append(a)
        return groups
    alone, mixed = build(), build()
    for a in alone:
        ops.compress_batch(ground_threshold=thr, acc=acc, **a)
    ops.compress_batch_mixed(mixed, ground_threshold=thr, acc=acc)
    torch.cuda.synchronize()
    for k, (a, m) in enumerate(zip(alone, mixed)):
        ba, bm = a["buf"], m["buf"]
        assert _beq(a["ground"].cpu().numpy(), m["ground"].cpu().numpy()), (seed, k)
        for f in ("ri", "seg", "cen_pix", "centers", "model", "counts", "nnz"):
            assert _beq(getattr(ba, f).cpu().numpy(), getattr(bm, f).cpu().numpy()), (seed, k, f)
        nz, qa, qm = ba.nnz.cpu().numpy(), ba.q16.cpu().numpy(), bm.q16.cpu().numpy()
        assert all(np.array_equal(qa[i, :nz[i]], qm[i, :nz[i]]) for i in range(ba.B)), (seed, k)
        if a["nonuniform"] is not None:
            assert np.array_equal(ba.salience.cpu().numpy(), bm.salience.cpu().numpy()), (seed, k)


def test_ground_fit_hand_over_equals_the_fit_alone(env):
    """Inside the fused call the band kernel hands the ground fit its candidate counts and, for images of whole 64-pixel words, a byte per pixel quad that
    says which pixels are candidates (the fit then walks set bits instead of testing every pixel).  Every variant -- bytes (P % 64 == 0), counts only
    (P % 4 == 0), the scalar write-out (odd P), no hand-over (a frame with a depth-0 point) -- must give the plane of rpcc_ground_ransac run alone on
    the same range image, bit for bit: more than 5000 candidates (systematic subsample), 800 .. 5000 (all kept), fewer than 800 (the whole cloud)."""
    torch, ops, orc, synth = env["torch"], env["ops"], env["orc"], env["synth"]
    for (H, W, vmax, vmin) in ((64, 2048, 2.0, -24.9), (32, 2250, 10.67, -30.67), (16, 1800, 15.0, -15.0),   # whole words
                               (20, 1250, 3.0, -25.0),      # P = 25000: quads, no whole words
                               (21, 1001, 3.0, -25.0)):     # P odd: scalar write-out
        g = orc.LidarGeom(H, W, 360.0, vmax, vmin)
        tm = ops.transform_map(g.H, g.W, g.horizontal_FOV, g.vertical_max, g.vertical_min)
        geom = ops.make_geom(g.H, g.W, g.horizontal_FOV, g.vertical_max, g.vertical_min)
        f0 = synth.make_frame(4400 + H, H, W, vmax_deg=vmax, vmin_deg=vmin).numpy()
        low = np.flatnonzero(f0[:, 2] < -1.5)
        frames = [f0,
                  np.concatenate([f0[f0[:, 2] >= -1.5], f0[low[:2500]]]),        # ~2500 candidates: all kept
                  f0[f0[:, 2] > -1.45],                                            # none: the whole cloud
                  np.concatenate([f0, np.zeros((1, 3), np.float32)])]              # a depth-0 point: the exact projection path, no hand-over
        n = len(frames)
        offs = np.zeros(n + 1, np.int64)
        offs[1:] = np.cumsum([f.shape[0] for f in frames])
        fid = _to(env, np.arange(n, dtype=np.int64) + 17)
        buf = ops.BatchBuffers(n, geom, 20, env["dev"])
        gms = torch.zeros((n, 4), dtype=torch.float64, device=env["dev"])
        ops.compress_batch(_to(env, np.concatenate(frames)), _to(env, offs), _to(env, tm), gms, buf, ground_seed=3, frame_ids=fid)
        alone, _ = ops.ground_ransac(buf.ri, _to(env, tm), seed=3, frame_ids=fid)
        assert _beq(gms.cpu().numpy(), alone.cpu().numpy()), (H, W)
        for i in (0, 1):
            ri = orc.project(frames[i], g)
            assert _beq(gms[i].cpu().numpy(), np.asarray(orc.ground_model(ri, tm, seed=3 + 17 + i), np.float64)), (H, W, i)


def test_ground_less_frames_are_fitted_chip_wide_to_the_same_plane(env):
    """A sweep with fewer than 800 ground candidates is fitted on EVERY pixel (segment_utils.py:105-106).  Inside the fused calls such a frame only draws
    its hypotheses in the per-frame launch; ground_wc_score_kernel scores them with workgroups all over the chip and ground_wc_refit_kernel refits
    (a whole-cloud fit inside one workgroup lasts 1 ms and the launch lasts as long as its slowest frame).  The planes must be those of rpcc_ground_ransac
    run alone (the one-workgroup form) and of the oracle, bit for bit: batches with one, several and only ground-less sweeps, the mixed-geometry call,
    and the ground stage issued twice (nothing left over from the first run)."""
    torch, ops, orc, synth = env["torch"], env["ops"], env["orc"], env["synth"]
    cases = {}
    for name, (H, W, vmax, vmin) in (("64", (64, 2048, 2.0, -24.9)), ("16", (16, 1800, 15.0, -15.0)), ("32", (32, 2250, 10.67, -30.67))):
        g = orc.LidarGeom(H, W, 360.0, vmax, vmin)
        tm = ops.transform_map(g.H, g.W, g.horizontal_FOV, g.vertical_max, g.vertical_min)
        geom = ops.make_geom(g.H, g.W, g.horizontal_FOV, g.vertical_max, g.vertical_min)
        full = [synth.make_frame(5200 + i, H, W, vmax_deg=vmax, vmin_deg=vmin).numpy() for i in range(6)]
        bare = [f[f[:, 2] > -1.45] for f in full]                      # no ground return at all
        few = [np.concatenate([b, f[f[:, 2] < -1.5][:500]]) for b, f in zip(bare, full)]   # 500 candidates: below the 800
        cases[name] = (g, geom, tm, full, bare, few)
    for name, pick in (("64", lambda full, bare, few: [full[0], bare[1], full[2], full[3], few[4], full[5]]),       # two of six
                       ("64", lambda full, bare, few: bare + few),                                                   # all twelve
                       ("16", lambda full, bare, few: [bare[0]] + full[1:]), ("32", lambda full, bare, few: [full[0], few[1], bare[2]])):
        g, geom, tm, full, bare, few = cases[name]
        frames = pick(full, bare, few)
        n = len(frames)
        offs = np.zeros(n + 1, np.int64)
        offs[1:] = np.cumsum([f.shape[0] for f in frames])
        fid = _to(env, np.arange(n, dtype=np.int64) + 40)
        buf = ops.BatchBuffers(n, geom, 20, env["dev"])
        gms = torch.zeros((n, 4), dtype=torch.float64, device=env["dev"])
        xyz, d_offs, d_tm = _to(env, np.concatenate(frames)), _to(env, offs), _to(env, tm)
        ops.compress_batch(xyz, d_offs, d_tm, gms, buf, ground_seed=5, frame_ids=fid)
        alone, _ = ops.ground_ransac(buf.ri, d_tm, seed=5, frame_ids=fid)
        assert _beq(gms.cpu().numpy(), alone.cpu().numpy()), (name, n)
        for i in range(min(n, 3)):
            assert _beq(gms[i].cpu().numpy(), np.asarray(orc.ground_model(orc.project(frames[i], g), tm, seed=5 + 40 + i), np.float64)), (name, i)
        # the ground stage again on the same buffers (twice), then the rest: the outputs of the single call
        seg0, q0, nz0 = buf.seg.clone(), buf.q16.clone(), buf.nnz.clone()
        gms2 = torch.zeros_like(gms)
        buf2 = ops.BatchBuffers(n, geom, 20, env["dev"])
        kw = dict(ground_seed=5, frame_ids=fid)
        ops.compress_batch_stages(ops.STAGE_PROJECT | ops.STAGE_GROUND, xyz, d_offs, d_tm, gms2, buf2, **kw)
        ops.compress_batch_stages(ops.STAGE_GROUND, xyz, d_offs, d_tm, gms2, buf2, **kw)
        ops.compress_batch_stages(127 & ~(ops.STAGE_PROJECT | ops.STAGE_GROUND), xyz, d_offs, d_tm, gms2, buf2, **kw)
        assert _beq(gms2.cpu().numpy(), gms.cpu().numpy()) and torch.equal(buf2.seg, seg0) and torch.equal(buf2.nnz, nz0), name
        assert all(torch.equal(buf2.q16[i, :int(nz0[i])], q0[i, :int(nz0[i])]) for i in range(n))
    # the mixed-geometry call: one launch over the groups' frames, ground-less sweeps in two of three groups
    groups, want = [], []
    for name, sel in (("64", lambda full, bare, few: [full[0], bare[1]]), ("16", lambda full, bare, few: [bare[0], few[1], full[2]]),
                      ("32", lambda full, bare, few: [full[0], full[1]])):
        g, geom, tm, full, bare, few = cases[name]
        frames = sel(full, bare, few)
        n = len(frames)
        offs = np.zeros(n + 1, np.int64)
        offs[1:] = np.cumsum([f.shape[0] for f in frames])
        fid = _to(env, np.arange(n, dtype=np.int64) + 7)
        buf = ops.BatchBuffers(n, geom, 20, env["dev"], general=True)
        groups.append(dict(xyz=_to(env, np.concatenate(frames)), offsets=_to(env, offs), tm=_to(env, tm), ground=torch.zeros((n, 4), dtype=torch.float64, device=env["dev"]),
                           buf=buf, ground_seed=9, frame_ids=fid))
        want.append((frames, g, tm))
    ops.compress_batch_mixed(groups)
    torch.cuda.synchronize()
    for gr, (frames, g, tm) in zip(groups, want):
        alone, _ = ops.ground_ransac(gr["buf"].ri, gr["tm"], seed=9, frame_ids=gr["frame_ids"])
        assert _beq(gr["ground"].cpu().numpy(), alone.cpu().numpy()), g.H
        assert _beq(gr["ground"][1].cpu().numpy(), np.asarray(orc.ground_model(orc.project(frames[1], g), tm, seed=9 + 7 + 1), np.float64)), g.H


def test_ground_less_frames_in_a_batch_of_more_than_1024(env):
    """The chip-wide scoring lists the marked frames 1024 at a time (WC_LIST): a batch of 1100 small sweeps in which every third one -- and a run of
    them across the 1024 boundary -- has no ground returns gives the planes of rpcc_ground_ransac alone, frame for frame."""
    torch, ops, orc, synth = env["torch"], env["ops"], env["orc"], env["synth"]
    H, W, vmax, vmin = 16, 512, 15.0, -15.0
    g = orc.LidarGeom(H, W, 360.0, vmax, vmin)
    tm = ops.transform_map(g.H, g.W, g.horizontal_FOV, g.vertical_max, g.vertical_min)
    geom = ops.make_geom(g.H, g.W, g.horizontal_FOV, g.vertical_max, g.vertical_min)
    base = [synth.make_frame(6100 + i, H, W, vmax_deg=vmax, vmin_deg=vmin).numpy() for i in range(12)]
    B = 1100
    frames = []
    for i in range(B):
        f = base[i % 12]
        frames.append(f[f[:, 2] > -1.45] if (i % 3 == 0 or 1015 <= i < 1040) else f)
    offs = np.zeros(B + 1, np.int64)
    offs[1:] = np.cumsum([f.shape[0] for f in frames])
    fid = _to(env, np.arange(B, dtype=np.int64) + 3)
    buf = ops.BatchBuffers(B, geom, 10, env["dev"])
    gms = torch.zeros((B, 4), dtype=torch.float64, device=env["dev"])
    d_tm = _to(env, tm)
    ops.compress_batch(_to(env, np.concatenate(frames)), _to(env, offs), d_tm, gms, buf, ground_seed=2, frame_ids=fid)
    alone, _ = ops.ground_ransac(buf.ri, d_tm, seed=2, frame_ids=fid)
    a, b = gms.cpu().numpy().view(np.uint64), alone.cpu().numpy().view(np.uint64)
    bad = np.flatnonzero((a != b).any(1))
    assert bad.size == 0, bad[:10]
    for i in (0, 1023, 1024, 1039, 1099):
        assert _beq(gms[i].cpu().numpy(), np.asarray(orc.ground_model(orc.project(frames[i], g), tm, seed=2 + 3 + i), np.float64)), i


def test_compress_batch_stages_equal_the_single_call(env):
    """rpcc_compress_batch_stages: the batch's stages issued one by one (and in two groups on two streams joined by an event) give the outputs of
    rpcc_compress_batch; all bits at once is the same call."""
    torch, ops, synth = env["torch"], env["ops"], env["synth"]
    g, geom, tm = _geom(env, "VelodyneVLP16")
    gd = env["orc"].GEOMS["VelodyneVLP16"]
    xyz, offs = synth.make_batch(range(880, 885), g.H, g.W, device=env["dev"], vmax_deg=gd["vmax_deg"], vmin_deg=gd["vmin_deg"])
    d_tm = _to(env, tm)
    def fresh():
        return ops.BatchBuffers(5, geom, 100, env["dev"]), torch.zeros((5, 4), dtype=torch.float64, device=env["dev"])
    ref, g0 = fresh()
    ops.compress_batch(xyz, offs, d_tm, g0, ref, ground_seed=3)
    one, g1 = fresh()
    for bit in range(7):
        ops.compress_batch_stages(1 << bit, xyz, offs, d_tm, g1, one, ground_seed=3)
    two, g2 = fresh()
    s2 = torch.cuda.Stream(device=env["dev"])
    ops.compress_batch_stages(ops.STAGE_PROJECT | ops.STAGE_GROUND | ops.STAGE_MASK, xyz, offs, d_tm, g2, two, ground_seed=3)
    e = torch.cuda.Event(); e.record()
    with torch.cuda.stream(s2):
        s2.wait_event(e)
        ops.compress_batch_stages(ops.STAGE_FPS | ops.STAGE_LABELS | ops.STAGE_PLANES | ops.STAGE_QUANTISE, xyz, offs, d_tm, g2, two, ground_seed=3)
    allb, g3 = fresh()
    ops.compress_batch_stages(127, xyz, offs, d_tm, g3, allb, ground_seed=3)
    torch.cuda.synchronize()
    for b, gg in ((one, g1), (two, g2), (allb, g3)):
        assert torch.equal(gg, g0) and torch.equal(b.seg, ref.seg) and torch.equal(b.nnz, ref.nnz) and torch.equal(b.cen_pix, ref.cen_pix)
        assert _beq(b.model.cpu().numpy()[:, :102], ref.model.cpu().numpy()[:, :102])
        for i in range(5):
            assert torch.equal(b.q16[i, :int(ref.nnz[i])], ref.q16[i, :int(ref.nnz[i])])

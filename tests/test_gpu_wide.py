"""-m gpu: cluster_num above 254 -- uint16 labels through the rpcc_*_wide entries (csrc/wide_kernels.h) -- against the CPU oracle, and the same
entries at cluster_num = 100 against the byte-label kernels (two independent implementations of one specification)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    import rpcc_amd  # noqa: F401
    from rpcc_amd import compress_utils, ops, pipeline, synth
    from rpcc_amd.tools.decompress import decode_frame
    from rpcc_amd.transformer import PCTransformer
    from oracle import oracle as orc
    return dict(torch=torch, ops=ops, synth=synth, orc=orc, pl=pipeline, cu=compress_utils, dec=decode_frame, T=PCTransformer, dev=torch.device("cuda:0"))


def _geom(env, name):
    orc, ops = env["orc"], env["ops"]
    gd = orc.GEOMS[name]
    g = orc.LidarGeom(**gd)
    tm = ops.transform_map(g.H, g.W, g.horizontal_FOV, g.vertical_max, g.vertical_min)
    return gd, g, ops.make_geom(g.H, g.W, g.horizontal_FOV, g.vertical_max, g.vertical_min), tm


def _batch(env, gd, g, ids, scene="default"):
    frames = [env["synth"].make_frame(i, g.H, g.W, vmax_deg=gd["vmax_deg"], vmin_deg=gd["vmin_deg"], scene=scene).numpy() for i in ids]
    offs = np.zeros(len(frames) + 1, np.int64)
    offs[1:] = np.cumsum([f.shape[0] for f in frames])
    to = lambda a: env["torch"].from_numpy(np.ascontiguousarray(a)).to(env["dev"])
    return frames, to(np.concatenate(frames)), to(offs)


@pytest.mark.parametrize("M,uniform,method", [(M, u, m) for M in (300, 1000) for (u, m) in ((True, "point"), (False, "point"), (True, "plane"), (False, "plane"))] +
                         [(1100, True, "point"), (1100, False, "point"), (1100, True, "plane"), (1100, False, "plane"), (1022, True, "point"), (1023, True, "point")])
def test_wide_batch_vs_oracle(env, M, uniform, method):
    """cluster_num = 300 / 1000 (labels up to 1001: uint16) on VLP-16 sweeps, all four framework / model combinations: fitted ground plane, FPS
    pixels, labels, model rows, salience levels and quantised integers equal the oracle's.  Up to 1022 clusters the batch runs the tuned
    assignment / histogram / label-order / quantiser kernels on uint16 labels (their label tables still fit LDS); 1023 and 1100 take the radix-sort
    kernels of wide_kernels.h."""
    torch, ops, orc, dev = env["torch"], env["ops"], env["orc"], env["dev"]
    gd, g, geom, tm = _geom(env, "VelodyneVLP16")
    ids = [7100, 7101, 7102]
    frames, xyz, offs = _batch(env, gd, g, ids)
    buf = ops.BatchBuffers(len(ids), geom, M, dev)
    assert buf.wide and buf.seg.dtype == torch.uint16
    gms = torch.zeros((len(ids), 4), dtype=torch.float64, device=dev)
    cfg = dict(orc.DEFAULT_CFG, cluster_num=M, plane_angle_threshold=75)
    nu = None if uniform else ops.nonuniform_cfg(0.04, cfg)
    fid = torch.as_tensor(np.asarray(ids, np.int64), device=dev)
    ops.compress_batch(xyz, offs, torch.from_numpy(tm).to(dev), gms, buf, ground_seed=5, frame_ids=fid, model_method=method, plane_seed=5, nonuniform=nu)
    torch.cuda.synchronize()
    seg, q16, nnz = buf.seg.cpu().numpy(), buf.q16.cpu().numpy(), buf.nnz.cpu().numpy()
    for i, f in enumerate(frames):
        gm = orc.ground_model(orc.project(f, g), tm, seed=5 + ids[i])
        assert np.array_equal(gms[i].cpu().numpy().view(np.uint64), np.asarray(gm, np.float64).view(np.uint64))
        o = orc.compress_frame(f, g, tm, gm, cfg, uniform=uniform, plane=None if method == "point" else dict(angle_deg=75, seed=5, frame=ids[i]))
        tag = (M, uniform, method, i)
        assert np.array_equal(buf.cen_pix[i].cpu().numpy(), o["fps_pix"]), tag
        assert int(o["seg_idx"].max()) > 255, "the case must need uint16 labels"
        bad = np.flatnonzero(seg[i].reshape(-1) != o["seg_idx"].reshape(-1))
        assert bad.size == 0, (tag, bad[:6], seg[i].reshape(-1)[bad[:6]], o["seg_idx"].reshape(-1)[bad[:6]])
        mp = np.asarray(o["model_param"]).astype(np.float32)
        assert np.array_equal(buf.model[i, :mp.shape[0]].cpu().numpy().view(np.uint32), mp.view(np.uint32)), tag
        assert np.array_equal(buf.counts[i, :mp.shape[0]].cpu().numpy(), np.bincount(o["seg_idx"].reshape(-1), minlength=mp.shape[0])), tag
        n = int(nnz[i])
        assert n == o["q"].shape[0] and np.array_equal(q16[i, :n], o["q"].astype(np.int16)), tag
        if not uniform:
            assert np.array_equal(buf.key_point_map[i].cpu().numpy(), o["key_point_map"].astype(np.uint8)), tag
            assert np.array_equal(buf.salience[i, :o["salience"].shape[0]].cpu().numpy(), o["salience"].astype(np.uint8)), tag


def test_mid_cluster_counts_with_the_bruteforce_fps_and_groundless_frames(env):
    """cluster_num = 300 through the fused plan (compress_batch_mid): the one-pass-per-centre FPS (RPCC_FPS_BRUTEFORCE) gives the same centres, labels
    and integers as the pruned one, and a batch that holds a sweep without ground returns (whole-cloud fit, scored chip-wide) equals the oracle."""
    torch, ops, orc, dev = env["torch"], env["ops"], env["orc"], env["dev"]
    gd, g, geom, tm = _geom(env, "VelodyneVLP16")
    ids = [7300, 7301, 7302]
    frames, xyz, offs = _batch(env, gd, g, ids)
    frames[1] = frames[1][frames[1][:, 2] > -1.45]                    # no ground returns
    offs_np = np.zeros(len(frames) + 1, np.int64)
    offs_np[1:] = np.cumsum([f.shape[0] for f in frames])
    xyz, offs = torch.from_numpy(np.concatenate(frames)).to(dev), torch.from_numpy(offs_np).to(dev)
    M = 300
    cfg = dict(orc.DEFAULT_CFG, cluster_num=M)
    fid = torch.as_tensor(np.asarray(ids, np.int64), device=dev)
    outs = []
    for brute in (False, True):
        buf = ops.BatchBuffers(len(ids), geom, M, dev)
        gms = torch.zeros((len(ids), 4), dtype=torch.float64, device=dev)
        ops.compress_batch(xyz, offs, torch.from_numpy(tm).to(dev), gms, buf, ground_seed=5, frame_ids=fid, fps_bruteforce=brute)
        torch.cuda.synchronize()
        outs.append((gms.cpu().numpy(), buf.cen_pix.cpu().numpy(), buf.seg.cpu().numpy(), buf.nnz.cpu().numpy(), buf.q16.cpu().numpy(), buf.model.cpu().numpy()))
    for u, v in zip(*outs):
        assert np.array_equal(u.view(np.uint8), v.view(np.uint8)) if u.dtype != np.int16 else all(
            np.array_equal(u[i, :outs[0][3][i]], v[i, :outs[0][3][i]]) for i in range(len(ids)))
    gm_d, pix, seg, nnz, q16, _ = outs[0]
    for i, f in enumerate(frames):
        gm = orc.ground_model(orc.project(f, g), tm, seed=5 + ids[i])
        assert np.array_equal(gm_d[i].view(np.uint64), np.asarray(gm, np.float64).view(np.uint64)), i
        o = orc.compress_frame(f, g, tm, gm, cfg)
        assert np.array_equal(pix[i], o["fps_pix"]) and np.array_equal(seg[i].reshape(-1), o["seg_idx"].reshape(-1)), i
        assert int(nnz[i]) == o["q"].shape[0] and np.array_equal(q16[i, :nnz[i]], o["q"].astype(np.int16)), i


@pytest.mark.parametrize("uniform,method,scene", [(True, "point", "default"), (False, "plane", "default"), (True, "point", "shell"),
                                                  (True, "point", "noise"), (True, "point", "corridor")])
def test_wide_kernels_equal_the_byte_label_kernels(env, uniform, method, scene):
    """The uint16 entries are a second, independent implementation of the same stages: at cluster_num = 100 on 64 x 2048 sweeps -- where the tuned
    byte-label kernels run as well -- both produce the same labels, model rows, counts, salience levels and quantised integers.  The adversarial
    scenes of the FPS study (all points equally far / independent ranges per pixel / a corridor) stress the centre screen of wide_assign_kernel:
    flat boxes with many near-ties, boxes that exclude nothing, very near and very far centres."""
    torch, ops, orc, dev = env["torch"], env["ops"], env["orc"], env["dev"]
    from rpcc_amd import _lib
    import ctypes as C
    gd, g, geom, tm = _geom(env, "Velodyne64E_2048")
    ids = [7200, 7201, 7202, 7203]
    frames, xyz, offs = _batch(env, gd, g, ids, scene)
    B, M = len(ids), 100
    d_tm = torch.from_numpy(tm).to(dev)
    fid = torch.as_tensor(np.asarray(ids, np.int64), device=dev)
    cfg = dict(orc.DEFAULT_CFG, plane_angle_threshold=75)
    nu = None if uniform else ops.nonuniform_cfg(0.04, cfg)
    ref = ops.BatchBuffers(B, geom, M, dev, general=True)
    g_ref = torch.zeros((B, 4), dtype=torch.float64, device=dev)
    ops.compress_batch(xyz, offs, d_tm, g_ref, ref, ground_seed=9, frame_ids=fid, model_method=method, plane_seed=9, nonuniform=nu)
    # the wide entry on the same batch: buffers as BatchBuffers would make them for a wide cluster count
    wide = ops.BatchBuffers(B, geom, M, dev, general=True)
    wide.seg = torch.empty((B, g.H, g.W), dtype=torch.uint16, device=dev)
    wide.ws = torch.empty(_lib.lib().rpcc_wide_workspace_bytes(B, g.H * g.W, M, int(xyz.shape[0])), dtype=torch.uint8, device=dev)
    g_w = torch.zeros((B, 4), dtype=torch.float64, device=dev)
    io = ops._batch_io(xyz, offs, d_tm, g_w, wide, 9, fid, False, None, method, 75, 9, nu, None, None)
    _lib.check(_lib.lib().rpcc_compress_batch_wide(C.byref(io), B, geom, M, 0.1, 0.04, _lib.ptr(wide.ws), _lib.stream()))
    torch.cuda.synchronize()
    assert np.array_equal(g_ref.cpu().numpy().view(np.uint64), g_w.cpu().numpy().view(np.uint64))
    assert np.array_equal(ref.cen_pix.cpu().numpy(), wide.cen_pix.cpu().numpy())
    assert np.array_equal(ref.seg.cpu().numpy().astype(np.uint16), wide.seg.cpu().numpy())
    assert np.array_equal(ref.counts.cpu().numpy(), wide.counts.cpu().numpy())
    assert np.array_equal(ref.nnz.cpu().numpy(), wide.nnz.cpu().numpy())
    assert np.array_equal(ref.model.cpu().numpy().view(np.uint32), wide.model.cpu().numpy().view(np.uint32))
    for i in range(B):
        n = int(ref.nnz[i])
        assert np.array_equal(ref.q16[i, :n].cpu().numpy(), wide.q16[i, :n].cpu().numpy()), i
    if not uniform:
        assert np.array_equal(ref.key_point_map.cpu().numpy(), wide.key_point_map.cpu().numpy())
        assert np.array_equal(ref.salience.cpu().numpy(), wide.salience.cpu().numpy())


@pytest.mark.parametrize("uniform,method", [(True, "point"), (False, "plane")])
def test_wide_front_end_round_trip(env, uniform, method):
    """pipeline.BatchCompressor(cluster_num=300): the .rpcc strings equal the oracle's container bytes (labels as uint16, as the reference writes
    them: utils/compress_utils.py:160), and tools/decompress.decode_frame recovers the labels and stays inside the error bound."""
    orc, dev = env["orc"], env["dev"]
    gd, g, geom, tm = _geom(env, "VelodyneVLP16")
    M, acc = 300, 0.02
    T = env["T"](dict(HORIZONTAL_FOV=gd["hfov_deg"], VERTICAL_ANGLE_MAX=gd["vmax_deg"], VERTICAL_ANGLE_MIN=gd["vmin_deg"], RANGE_IMAGE_HEIGHT=g.H, RANGE_IMAGE_WIDTH=g.W))
    frames = [env["synth"].make_frame(7300 + i, g.H, g.W, vmax_deg=gd["vmax_deg"], vmin_deg=gd["vmin_deg"]).numpy() for i in range(3)]
    cfg = dict(orc.DEFAULT_CFG, cluster_num=M, accuracy=acc, plane_angle_threshold=75)
    bc = env["pl"].BatchCompressor(T, cluster_num=M, accuracy=acc, uniform=uniform, model_method=method, compressor_cfg=cfg, seed=13)
    blobs = bc.compress(frames)
    lacc = np.array([2 * acc] * 4) + np.array([0, 0.02, 0.04, 0.06])
    for b, f in enumerate(frames):
        ri = orc.project(f, g)
        gm = orc.ground_model(ri, tm, seed=13 + b)
        o = orc.compress_frame(f, g, tm, gm, cfg, uniform=uniform, plane=None if method == "point" else dict(angle_deg=75, seed=13, frame=b))
        od = orc.pack_payload(o["model_param"], o["seg_idx"], None if uniform else o["salience"], o["q"])
        assert blobs[b] == orc.bitstream_bytes(od, uniform=uniform), (uniform, method, b)
        rec, pc, seg_rec = env["dec"](env["cu"].unpack_bitstream(blobs[b], uniform=uniform), env["cu"].BasicCompressor(method_name="bzip2"), T, M, 2 * acc,
                                      lacc, uniform=uniform)
        assert np.array_equal(np.asarray(seg_rec).astype(np.int64), o["seg_idx"]), (uniform, method, b)
        err = np.abs(rec - ri)[ri != 0]
        assert err.max() <= (acc if uniform else 2 * acc + 0.06) + 1e-5


def test_wide_cli_roundtrip(env, tmp_path):
    """tools/compress.py / decompress.py / compress_datalist.py with --cluster_num 300 (a legal value of the reference's YAML, cfgs/compressor.yaml:22):
    the single-frame tool takes the batch front-end for it, the datalist tool writes the same bytes, the decoder returns every non-empty pixel."""
    import os
    from rpcc_amd.tools import compress as tc, compress_datalist as tdl, decompress as td
    orc = env["orc"]
    gd = orc.GEOMS["VelodyneVLP16"]
    f = env["synth"].make_frame(7400, gd["H"], gd["W"], vmax_deg=gd["vmax_deg"], vmin_deg=gd["vmin_deg"]).numpy()
    src = tmp_path / "sweep.bin"
    np.concatenate((f, np.zeros((f.shape[0], 1), np.float32)), 1).astype(np.float32).tofile(src)
    for extra in ([], ["--nonuniform"]):
        out = tmp_path / ("sweep%s.rpcc" % ("_n" if extra else ""))
        common = ["--lidar", "VelodyneVLP16", "--cluster_num", "300"] + extra
        tc.compress(tc.make_parser().parse_args(["--input", str(src), "--output", str(out), "--eval"] + common))   # --eval raises beyond the bound
        rec = tmp_path / "rec.bin"
        td.decompress(tc.make_parser().parse_args(["--input", str(out), "--output", str(rec)] + common))
        g = orc.LidarGeom(**gd)
        assert np.fromfile(rec, dtype=np.float32).reshape(-1, 4).shape[0] == int((orc.project(f, g) != 0).sum())
        lst = tmp_path / "list.txt"
        lst.write_text(str(src) + "\n")
        od = tmp_path / ("out%d" % len(extra))
        tdl.compress(tc.make_parser(datalist=True).parse_args(["--datalist", str(lst), "--output_dir", str(od), "--batch", "2"] + common))
        assert open(tdl.output_path_for(str(od), str(src)), "rb").read() == open(out, "rb").read()


def test_wide_digests_of_the_reference(env):
    """The two VLP-16 sweeps the genuine reference compressed with cluster_num = 300 (tests/golden/manifest_sha.json "wide"): labels, model rows,
    quantised integers and the .rpcc bytes of the batch front-end have the reference's digests."""
    import hashlib
    import json
    import os
    orc = env["orc"]
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    w = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "manifest_sha.json")))["wide"]
    gd, g, geom, tm = _geom(env, w["geom"])
    T = env["T"](dict(HORIZONTAL_FOV=gd["hfov_deg"], VERTICAL_ANGLE_MAX=gd["vmax_deg"], VERTICAL_ANGLE_MIN=gd["vmin_deg"], RANGE_IMAGE_HEIGHT=g.H, RANGE_IMAGE_WIDTH=g.W))
    rows = w["rows"]
    frames = [env["synth"].make_frame(r["frame"], g.H, g.W, vmax_deg=gd["vmax_deg"], vmin_deg=gd["vmin_deg"]).numpy() for r in rows]
    bc = env["pl"].BatchCompressor(T, cluster_num=rows[0]["cluster_num"], accuracy=0.02)
    blobs = bc.compress(frames, ground=np.array([r["ground_model"] for r in rows]))
    buf = bc._buf
    for b, r in enumerate(rows):
        s = r["sha"]
        assert sha(frames[b]) == s["xyz"]
        assert sha(buf.seg[b].cpu().numpy()) == s["seg_idx"] and sha(buf.model[b, :r["labels"]].cpu().numpy()) == s["model_param"]
        n = int(buf.nnz[b])
        assert n == r["nnz"] and sha(buf.q16[b, :n].cpu().numpy()) == s["q"]
        assert len(blobs[b]) == r["rpcc_bytes"] and hashlib.sha256(blobs[b]).hexdigest() == s["rpcc"]


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_wide_vs_oracle(env, seed):
    """Randomised breadth for the uint16-label path: image shape (odd widths too), fields of view, cluster count 255 .. 700, accuracy, framework and model
    drawn per seed; labels, model rows, salience levels, quantised integers and the .rpcc bytes equal the oracle's."""
    orc = env["orc"]
    rng = np.random.default_rng(9100 + seed)
    H, W = int(rng.integers(8, 41)), int(rng.integers(300, 1500))
    vmax, vmin = float(rng.uniform(1.0, 16.0)), float(-rng.uniform(10.0, 31.0))
    M = int(rng.integers(255, 701))
    accuracy = float(rng.choice([0.01, 0.02, 0.05]))
    uniform, method = bool(rng.integers(0, 2)), ("plane" if rng.integers(0, 2) else "point")
    T = env["T"](dict(HORIZONTAL_FOV=360, VERTICAL_ANGLE_MAX=vmax, VERTICAL_ANGLE_MIN=vmin, RANGE_IMAGE_HEIGHT=H, RANGE_IMAGE_WIDTH=W))
    g = orc.LidarGeom(H, W, 360, vmax, vmin)
    tm = orc.transform_map(g)
    frames = [env["synth"].make_frame(9300 + 10 * seed + i, H, W, vmax_deg=vmax, vmin_deg=vmin).numpy() for i in range(2)]
    cfg = dict(orc.DEFAULT_CFG, accuracy=accuracy, cluster_num=M, plane_angle_threshold=75)
    bc = env["pl"].BatchCompressor(T, cluster_num=M, accuracy=accuracy, uniform=uniform, model_method=method, compressor_cfg=cfg, seed=31)
    blobs = bc.compress(frames)
    buf = bc._buf
    tag = (seed, H, W, M, uniform, method)
    compared = 0
    for b, f in enumerate(frames):
        ri = orc.project(f, g)
        gm = orc.ground_model(ri, tm, seed=31 + b)
        o = orc.compress_frame(f, g, tm, gm, cfg, uniform=uniform, plane=None if method == "point" else dict(angle_deg=75, seed=31, frame=b))
        if len(set(o["fps_pix"].tolist())) < M:      # fewer candidates than clusters: the centre list repeats (degenerate input)
            continue
        assert np.array_equal(buf.seg[b].cpu().numpy().astype(np.int64), o["seg_idx"]), tag
        mp = np.asarray(o["model_param"]).astype(np.float32)
        assert np.array_equal(buf.model[b, :mp.shape[0]].cpu().numpy().view(np.uint32), mp.view(np.uint32)), tag
        n = int(buf.nnz[b])
        assert n == o["q"].shape[0] and np.array_equal(buf.q16[b, :n].cpu().numpy(), o["q"].astype(np.int16)), tag
        od = orc.pack_payload(o["model_param"], o["seg_idx"], None if uniform else o["salience"], o["q"])
        assert blobs[b] == orc.bitstream_bytes(od, uniform=uniform), tag
        compared += 1
    assert compared >= 1, tag

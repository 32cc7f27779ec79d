"""-m gpu: the Python front-end that mirrors the reference's classes and tools,
driven the way tools/compress.py drives the reference, must produce the reference's .rpcc bytes."""
import json
import os
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
MAN = json.load(open(os.path.join(HERE, "golden", "manifest.json")))
LIDAR = {"Velodyne64E": "Velodyne64E", "Velodyne64E_2048": "Velodyne64E_2048", "Velodyne32E": "Velodyne32E",
         "VelodyneVLP16": "VelodyneVLP16"}
SHA = json.load(open(os.path.join(HERE, "golden", "manifest_sha.json")))


def _dataset(fe, geom):
    """The registry entry of a golden geometry: a lidar type, or -- for the 80 x 2000 table -- the dataset that uses it
    (dataset/__init__.py:21 'KITTI_test')."""
    if geom == "Velodyne64E_unofficial":
        return fe.ds.build_dataset(dataset_name="KITTI_test")
    return fe.ds.build_dataset(lidar_type=LIDAR[geom])


@pytest.fixture(scope="module")
def fe():
    import torch
    assert torch.cuda.is_available()
    import rpcc_amd  # noqa: F401
    from rpcc_amd import compress_utils, dataset, pipeline, segment_utils
    from rpcc_amd.tools import compress as tool_c, decompress as tool_d
    return types.SimpleNamespace(torch=torch, cu=compress_utils, ds=dataset, pl=pipeline, su=segment_utils, tc=tool_c,
                                 td=tool_d)


@pytest.mark.parametrize("case", sorted(MAN["cases"]))
def test_reference_style_driver_matches_golden_bytes(fe, case):
    """The body of tools/compress.py:93-135 written against the mirror classes, ground model injected
    through PointCloudSegment.ransac_plane_segmentation like a user of the reference would."""
    c = MAN["cases"][case]
    z = np.load(os.path.join(HERE, "golden", case + ".npz"))
    ds = _dataset(fe, c["geom"])
    T = ds.PCTransformer
    ri = np.expand_dims(T.point_cloud_to_range_image(z["xyz"]), -1)
    pc = T.range_image_to_point_cloud(ri)
    gm = z["ground_model"]
    fe.su.PointCloudSegment.ransac_plane_segmentation = staticmethod(lambda pts, *a, **k: (None, gm))
    try:
        seg_cfg = {"segment_method": "FPS", "ground_vertical_threshold": 0.1, "cluster_num": 100, "DBSCAN_eps": 1.5}
        ps = fe.su.PointCloudSegment(ds.transform_map)
        seg_idx, ground_model = ps.segment(pc, ri, seg_cfg, cpu=True)
    finally:
        fe.su.PointCloudSegment.ransac_plane_segmentation = None
    assert seg_idx.dtype == np.int64 and np.array_equal(seg_idx.astype(np.uint8), z["seg_idx"])
    cluster_models = ps.cluster_modeling(pc, ri, seg_idx, {"model_method": "point", "angle_threshold": 75})
    model_param = np.concatenate((ground_model.reshape(1, 4), cluster_models), 0)
    assert np.array_equal(model_param.astype(np.float32).view(np.uint32), z["model_param"].astype(np.float32).view(np.uint32))
    pred = ps.intra_predict(seg_idx, model_param)
    residual = ri - pred
    QM = fe.cu.QuantizationModule(0.04)
    q, sal, kp = QM.quantize_residual(residual, seg_idx, pc, ri)
    assert sal is None and np.array_equal(q.astype(np.int16), z["q_uniform"])
    bc = fe.cu.BasicCompressor(method_name="bzip2")
    od, cd = fe.cu.compress_point_cloud(bc, model_param, seg_idx, sal, q, full=False)
    blob = fe.cu.pack_bitstream(cd, uniform=True)
    assert blob == z["rpcc"].tobytes()
    # non-uniform quantiser through the same seam
    QN = fe.cu.QuantizationModule(0.04, uniform=False)
    qn, saln, kpn = QN.quantize_residual(residual, seg_idx, pc, ri)
    assert np.array_equal(qn.astype(np.int16), z["q_nonuniform"]) and np.array_equal(saln.astype(np.uint8), z["salience"])
    assert np.array_equal(kpn.astype(np.uint8), z["key_point_map"])
    # decoder mirror: read back, dequantise, reconstruct
    rq, seg2, sal2, pp = fe.cu.decompress_point_cloud(fe.cu.unpack_bitstream(blob), bc, model_param.shape[0], T.H, T.W)
    assert np.array_equal(seg2, seg_idx) and np.array_equal(rq, z["q_uniform"])
    rec = ps.intra_predict(seg2, pp) + QM.dequantize_residual(rq, seg2)
    err = np.abs(rec - ri)[ri != 0]
    assert err.max() <= 0.02 + 1e-5


def test_batch_compressor_matches_golden_bytes(fe):
    """pipeline.BatchCompressor (the datalist tool's engine): several frames per call, injected ground."""
    z = np.load(os.path.join(HERE, "golden", "synth_64x2048.npz"))
    ds = fe.ds.build_dataset(lidar_type="Velodyne64E_2048")
    bc = fe.pl.BatchCompressor(ds.PCTransformer)
    blobs = bc.compress([z["xyz"], z["xyz"][::-1].copy(), z["xyz"]], ground=np.tile(z["ground_model"], (3, 1)))
    for b in blobs:
        assert b == z["rpcc"].tobytes()       # projection is order independent: the reversed frame too


def test_cli_roundtrip(fe, tmp_path):
    """tools/compress.py + tools/decompress.py with the reference's flags (run in-process), uniform and
    --nonuniform; compress_datalist.py on a 3-entry datalist."""
    z = np.load(os.path.join(HERE, "golden", "example_64E.npz"))
    src = tmp_path / "frame.bin"
    np.concatenate((z["xyz"], np.zeros((z["xyz"].shape[0], 1), np.float32)), 1).astype(np.float32).tofile(src)
    for extra in ([], ["--nonuniform"]):
        out = tmp_path / ("frame%s.rpcc" % ("_n" if extra else ""))
        a = fe.tc.make_parser().parse_args(["--input", str(src), "--output", str(out), "--lidar", "Velodyne64E", "--eval"] + extra)
        fe.tc.compress(a)          # --eval raises if the reconstruction bound is violated
        assert 20000 < os.path.getsize(out) < 60000
        rec = tmp_path / "rec.bin"
        d = fe.tc.make_parser().parse_args(["--input", str(out), "--output", str(rec), "--lidar", "Velodyne64E"] + extra)
        fe.td.decompress(d)
        pts = np.fromfile(rec, dtype=np.float32).reshape(-1, 4)
        assert pts.shape[0] == MAN["cases"]["example_64E"]["nnz"]
    from rpcc_amd.tools import compress_datalist as tdl
    lst = tmp_path / "list.txt"
    lst.write_text("\n".join([str(src)] * 3) + "\n")
    a = fe.tc.make_parser(datalist=True).parse_args(["--datalist", str(lst), "--output_dir", str(tmp_path / "out"),
                                                     "--lidar", "Velodyne64E", "--batch", "2"])
    tdl.compress(a)
    produced = tdl.output_path_for(str(tmp_path / "out"), str(src))
    assert os.path.getsize(produced) > 20000


def test_datalist_gather_writes_the_same_files(fe, tmp_path):
    """tools/compress_datalist.py --gather (the north star's "RCCL only for the final gather of compressed bitstreams"): the
    .rpcc bytes go through sharding.RoundGather over a single-rank RCCL group and rank 0 writes them -- the files equal the
    default mode's (every rank writes its own), with rounds shorter than the datalist."""
    from rpcc_amd import synth
    from rpcc_amd.tools import compress_datalist as tdl
    from oracle import oracle as orc
    gd = orc.GEOMS["VelodyneVLP16"]
    names = []
    for i in range(5):
        f = synth.make_frame(40 + i, gd["H"], gd["W"], vmax_deg=gd["vmax_deg"], vmin_deg=gd["vmin_deg"]).numpy()
        src = tmp_path / ("sweep_%03d.bin" % i)
        np.concatenate((f, np.zeros((f.shape[0], 1), np.float32)), 1).astype(np.float32).tofile(src)
        names.append(str(src))
    lst = tmp_path / "list.txt"
    lst.write_text("\n".join(names) + "\n")
    base = ["--datalist", str(lst), "--lidar", "VelodyneVLP16", "--batch", "2"]
    tdl.compress(fe.tc.make_parser(datalist=True).parse_args(base + ["--output_dir", str(tmp_path / "a")]))
    old = {k: os.environ.get(k) for k in ("MASTER_PORT",)}
    os.environ["MASTER_PORT"] = "29571"
    try:
        tdl.compress(fe.tc.make_parser(datalist=True).parse_args(base + ["--output_dir", str(tmp_path / "b"), "--gather", "--gather-round", "2"]))
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    for n in names:
        a, b = tdl.output_path_for(str(tmp_path / "a"), n), tdl.output_path_for(str(tmp_path / "b"), n)
        assert os.path.getsize(a) > 1000 and open(a, "rb").read() == open(b, "rb").read(), n


def test_streaming_loader_rows_ingest(fe, tmp_path):
    """ingest="rows": the sweeps as stored (float32 rows x, y, z, intensity; dataset/dataset.py:48-50) -- [N,4] arrays copied whole,
    or .bin PATHS read straight into the pinned slot -- go to the device unsliced (rpcc_batch_io.point_stride_bytes = 16).  The
    .rpcc bytes equal those of the default ingest (host-side [:, :3] slice); a slot that is too small grows; the datalist tool
    picks the mode by itself for a datalist of .bin files."""
    from oracle import oracle as orc
    from rpcc_amd import synth
    from rpcc_amd.loader import StreamingCompressor
    gd = orc.GEOMS["VelodyneVLP16"]
    ds = fe.ds.build_dataset(lidar_type="VelodyneVLP16")
    frames = [synth.make_frame(1700 + i, gd["H"], gd["W"], vmax_deg=gd["vmax_deg"], vmin_deg=gd["vmin_deg"]).numpy() for i in range(7)]
    frames[2] = np.zeros((0, 3), np.float32)
    rng = np.random.default_rng(4)
    rows = [np.concatenate([f, rng.random((f.shape[0], 1), dtype=np.float32)], 1) for f in frames]
    paths = []
    for i, r in enumerate(rows):
        paths.append(str(tmp_path / ("%06d.bin" % i)))
        r.tofile(paths[-1])
    ids = [40 + 3 * i for i in range(len(frames))]
    bc = fe.pl.BatchCompressor(ds.PCTransformer, accuracy=0.02, seed=9)
    want = {}
    StreamingCompressor(bc, batch=3, depth=2, workers=2).run(((frames[s:s + 3], ids[s:s + 3]) for s in range(0, 7, 3)),
                                                             sink=lambda k, r: want.__setitem__(k, r))
    for src in (rows, paths):
        sc = StreamingCompressor(bc, batch=3, depth=2, workers=2, ingest="rows", points_per_frame=64)   # (undersized on purpose)
        got = {}
        n = sc.run(((src[s:s + 3], ids[s:s + 3]) for s in range(0, 7, 3)), sink=lambda k, r: got.__setitem__(k, r))
        assert n == 7 and sc.grown >= 1 and sc.slots[0].xyz_dev.shape[1] == 4
        assert got == want
    # the datalist tool: .bin datalist -> rows by itself; the files equal those of --ingest xyz
    dl = tmp_path / "list.txt"
    dl.write_text("\n".join(paths) + "\n")
    from rpcc_amd.tools import compress_datalist as tdl
    outs = {}
    for mode in ("auto", "xyz"):
        od = tmp_path / ("out_" + mode)
        tdl.compress(fe.tc.make_parser(datalist=True).parse_args(["--datalist", str(dl), "--output_dir", str(od), "--lidar", "VelodyneVLP16",
                                                                  "--batch", "4", "--ingest", mode]))
        outs[mode] = {os.path.basename(f): open(os.path.join(d, f), "rb").read() for d, _, fs in os.walk(od) for f in fs}
    assert len(outs["auto"]) == 7 and outs["auto"] == outs["xyz"]


def test_streaming_loader_grows_its_slots(fe):
    """A batch with more points than one per pixel and frame (dense / dual-return sweeps): the staging slot grows instead of
    raising, and the bytes equal BatchCompressor.compress (which sizes its buffers from the data)."""
    from oracle import oracle as orc
    from rpcc_amd import synth
    from rpcc_amd.loader import StreamingCompressor
    gd = orc.GEOMS["VelodyneVLP16"]
    ds = fe.ds.build_dataset(lidar_type="VelodyneVLP16")
    base = [synth.make_frame(820 + i, gd["H"], gd["W"], vmax_deg=gd["vmax_deg"], vmin_deg=gd["vmin_deg"]).numpy() for i in range(4)]
    rng = np.random.default_rng(3)
    dense = [np.concatenate([f, f * np.float32(1.01), f[rng.permutation(f.shape[0])] * np.float32(0.99)]) for f in base]   # ~2.4 points per pixel
    assert all(f.shape[0] > gd["H"] * gd["W"] for f in dense)
    bc = fe.pl.BatchCompressor(ds.PCTransformer, accuracy=0.02, seed=2)
    want = bc.compress(dense[:2], frame_ids=[5, 6]) + bc.compress(dense[2:], frame_ids=[7, 8])
    sc = StreamingCompressor(bc, batch=2, depth=2, workers=2)
    got = {}
    n = sc.run(((dense[s:s + 2], [5 + s, 6 + s]) for s in (0, 2)), sink=lambda k, r: got.__setitem__(k, r))
    assert n == 4 and sc.grown >= 1
    assert [b for k in sorted(got) for b in got[k]] == want


@pytest.mark.parametrize("accuracy", [0.01, 0.02, 0.05])
@pytest.mark.parametrize("lidar,geom", [("Velodyne64E", "Velodyne64E"), ("Velodyne32E", "Velodyne32E"), ("VelodyneVLP16", "VelodyneVLP16")])
def test_nonuniform_plane_batches_mixed_lidars(fe, lidar, geom, accuracy):
    """BASELINE configs[2]/[4]: non-uniform framework + plane model, one batch per lidar geometry, accuracy
    sweep.  The batch path must equal the oracle run stage by stage with the same (fitted) models, and the
    .rpcc must decode within the non-uniform error bound (tools/compress.py:179-181)."""
    from oracle import oracle as orc
    from rpcc_amd import synth
    from rpcc_amd.tools.decompress import decode_frame
    gd = orc.GEOMS[geom]
    g = orc.LidarGeom(**gd)
    tm = orc.transform_map(g)
    ds = fe.ds.build_dataset(lidar_type=lidar)
    frames = [synth.make_frame(300 + i, g.H, g.W, vmax_deg=gd["vmax_deg"], vmin_deg=gd["vmin_deg"]).numpy() for i in range(3)]
    cfg = dict(orc.DEFAULT_CFG, accuracy=accuracy, plane_angle_threshold=75)
    bc = fe.pl.BatchCompressor(ds.PCTransformer, accuracy=accuracy, uniform=False, model_method="plane",
                               compressor_cfg=cfg, seed=11)
    blobs = bc.compress(frames)
    buf = bc._buf
    step = accuracy * 2
    lacc = np.array([step] * 4) + np.array([0, 0.02, 0.04, 0.06])
    for b, f in enumerate(frames):
        ri = orc.project(f, g)
        gm = orc.ground_model(ri, tm, seed=11 + b)
        s = orc.segment(ri, tm, gm, cfg)
        seg = s["seg_idx"]
        assert np.array_equal(buf.seg[b].cpu().numpy(), seg.astype(np.uint8))
        mp = np.concatenate((gm.reshape(1, 4), orc.cluster_modeling_plane(s["pc"], ri, seg, tm, 75, 11, b)), 0)
        nrow = mp.shape[0]
        assert np.array_equal(buf.model[b, :nrow].cpu().numpy().view(np.uint32), mp.astype(np.float32).view(np.uint32))
        pred = orc.intra_predict(seg, mp, tm)
        _, kp = orc.extract_features_with_segment(ri, seg)
        q, sal = orc.nonuniform_quantize(seg, ri.reshape(g.H, g.W, 1) - pred, kp, np.array([30, 10, 3, 0]), lacc, 2)
        n = int(buf.nnz[b])
        assert n == q.shape[0] and np.array_equal(buf.q16[b, :n].cpu().numpy(), q.astype(np.int16))
        od = orc.pack_payload(mp, seg, sal, q)
        assert blobs[b] == orc.bitstream_bytes(od, uniform=False)
        rec, pc, seg_rec = decode_frame(fe.cu.unpack_bitstream(blobs[b], uniform=False), fe.cu.BasicCompressor(method_name="bzip2"),
                                        ds.PCTransformer, 100, step, lacc, uniform=False)
        assert np.array_equal(seg_rec, seg.astype(np.uint8))
        err = np.abs(rec - ri)[ri != 0]
        assert err.max() <= step + 0.06 + 1e-5


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("RPCC_FUZZ_SEEDS_GENERAL", "12")))))
def test_fuzz_general_path_vs_oracle(fe, seed):
    """Randomised breadth for the stage-by-stage batch path (configs[2]/[4] shape): image shape, fields of view, cluster
    count, accuracy and framework / model combination drawn per seed; segmentation, plane or point models, key points +
    salience + quantised integers, the .rpcc bytes and the decoded labels / error bound equal the oracle's."""
    from oracle import oracle as orc
    from rpcc_amd import synth
    from rpcc_amd.tools.decompress import decode_frame
    from rpcc_amd.transformer import PCTransformer
    rng = np.random.default_rng(5000 + seed)
    H, W = int(rng.integers(6, 41)), int(rng.integers(200, 1600))
    vmax, vmin = float(rng.uniform(1.0, 16.0)), float(-rng.uniform(10.0, 31.0))
    M = int(rng.integers(5, 61))
    accuracy = float(rng.choice([0.01, 0.02, 0.05]))
    uniform = bool(rng.integers(0, 2))
    method = "plane" if (rng.integers(0, 3) > 0 or uniform) else "point"   # uniform + point is the fused entry's test
    angle = float(rng.choice([75, 75, 40]))
    T = PCTransformer(dict(HORIZONTAL_FOV=360, VERTICAL_ANGLE_MAX=vmax, VERTICAL_ANGLE_MIN=vmin, RANGE_IMAGE_HEIGHT=H,
                           RANGE_IMAGE_WIDTH=W))
    g = orc.LidarGeom(H, W, 360, vmax, vmin)
    tm = orc.transform_map(g)
    assert np.array_equal(T.transform_map, tm)
    frames = [synth.make_frame(8000 + 10 * seed + i, H, W, vmax_deg=vmax, vmin_deg=vmin).numpy() for i in range(2)]
    cfg = dict(orc.DEFAULT_CFG, accuracy=accuracy, cluster_num=M, plane_angle_threshold=angle)
    bc = fe.pl.BatchCompressor(T, cluster_num=M, accuracy=accuracy, uniform=uniform, model_method=method, compressor_cfg=cfg,
                               seed=21)
    blobs = bc.compress(frames)
    buf = bc._buf
    step = accuracy * 2
    lacc = np.array([step] * 4) + np.array([0, 0.02, 0.04, 0.06])
    tag = (seed, H, W, M, uniform, method)
    compared = 0
    for b, f in enumerate(frames):
        ri = orc.project(f, g)
        gm = orc.ground_model(ri, tm, seed=21 + b)
        s = orc.segment(ri, tm, gm, cfg)
        seg = s["seg_idx"]
        if len(set(s["fps_pix"].tolist())) < M:
            continue
        assert np.array_equal(buf.seg[b].cpu().numpy(), seg.astype(np.uint8)), tag
        if method == "plane":
            mp = np.concatenate((gm.reshape(1, 4), orc.cluster_modeling_plane(s["pc"], ri, seg, tm, angle, 21, b)), 0)
        else:
            mp = orc.point_model_param(ri, seg, gm)
        nrow = mp.shape[0]
        assert np.array_equal(buf.model[b, :nrow].cpu().numpy().view(np.uint32), mp.astype(np.float32).view(np.uint32)), tag
        pred = orc.intra_predict(seg, mp.astype(np.float32), tm)
        res = ri.reshape(H, W, 1) - pred
        if uniform:
            q, sal = orc.uniform_quantize(seg.astype(np.int32), res, step), None
        else:
            _, kp = orc.extract_features_with_segment(ri, seg)
            q, sal = orc.nonuniform_quantize(seg, res, kp, np.array([30, 10, 3, 0]), lacc, 2)
        n = int(buf.nnz[b])
        assert n == q.shape[0] and np.array_equal(buf.q16[b, :n].cpu().numpy(), q.astype(np.int16)), tag
        od = orc.pack_payload(mp, seg, sal, q)
        assert blobs[b] == orc.bitstream_bytes(od, uniform=uniform), tag
        rec, pc, seg_rec = decode_frame(fe.cu.unpack_bitstream(blobs[b], uniform=uniform), fe.cu.BasicCompressor(method_name="bzip2"),
                                        T, M, step, lacc, uniform=uniform)
        assert np.array_equal(seg_rec, seg.astype(np.uint8)), tag
        err = np.abs(rec - ri)[ri != 0]
        assert err.max() <= (step / 2 if uniform else step + 0.06) + 1e-5, tag
        compared += 1
    assert compared >= 1, tag


def test_mixed_lidar_batch(fe):
    """configs[4]: one call with sweeps of three lidar geometries interleaved (variable H x W), non-uniform framework + plane
    model: every frame's .rpcc equals what the single-geometry batch path gives for it (which the tests above pin to the oracle),
    whatever the grouping and the order."""
    from oracle import oracle as orc
    from rpcc_amd import synth
    names = ["VelodyneVLP16", "Velodyne64E", "Velodyne32E", "VelodyneVLP16", "Velodyne32E", "VelodyneVLP16", "Velodyne64E"]
    T = {n: fe.ds.build_dataset(lidar_type=n).PCTransformer for n in set(names)}
    frames = []
    for i, n in enumerate(names):
        gd = orc.GEOMS[n]
        frames.append(synth.make_frame(4000 + i, gd["H"], gd["W"], vmax_deg=gd["vmax_deg"], vmin_deg=gd["vmin_deg"]).numpy())
    kw = dict(accuracy=0.02, uniform=False, model_method="plane", seed=3)
    mixed = fe.pl.MixedBatchCompressor(T, **kw).compress(frames, names)
    assert all(isinstance(b, bytes) and len(b) > 1000 for b in mixed)
    for n in set(names):
        idx = [i for i, m in enumerate(names) if m == n]
        single = fe.pl.BatchCompressor(T[n], **kw).compress([frames[i] for i in idx])
        for i, blob in zip(idx, single):
            assert mixed[i] == blob, (n, i)


@pytest.mark.parametrize("geom", sorted(SHA["cases"]))
def test_breadth_digests_batch_front_end(fe, geom):
    """The eight seeded sweeps per geometry the genuine reference was run on (tests/golden/manifest_sha.json; every shipped lidar YAML
    incl. KITTI_test's 80 x 2000, and 64 x 2048) as ONE fused batch per geometry and framework: range images, labels, model rows,
    quantised integers, key points, salience levels and the .rpcc bytes have the reference's digests."""
    import hashlib
    from oracle import oracle as orc
    from rpcc_amd import synth
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    gd = orc.GEOMS[geom]
    T = _dataset(fe, geom).PCTransformer
    rows = SHA["cases"][geom]
    frames = [synth.make_frame(r["frame"], gd["H"], gd["W"], vmax_deg=gd["vmax_deg"], vmin_deg=gd["vmin_deg"]).numpy() for r in rows]
    for f, r in zip(frames, rows):
        assert sha(f) == r["sha"]["xyz"], "the synthetic input is not the one the fixture was made from"
    gms = np.array([r["ground_model"] for r in rows])
    bc = fe.pl.BatchCompressor(T, accuracy=0.02)
    blobs = bc.compress(frames, ground=gms)
    buf = bc._buf
    for b, r in enumerate(rows):
        s, tag = r["sha"], (geom, r["frame"])
        assert sha(buf.ri[b].cpu().numpy()) == s["ri"], tag
        assert sha(buf.seg[b].cpu().numpy()) == s["seg_idx"], tag
        assert sha(buf.model[b, :r["labels"]].cpu().numpy()) == s["model_param"], tag
        n = int(buf.nnz[b])
        assert n == r["nnz"] and sha(buf.q16[b, :n].cpu().numpy()) == s["q"], tag
        assert len(blobs[b]) == r["rpcc_bytes"] and hashlib.sha256(blobs[b]).hexdigest() == s["rpcc"], tag
    bn = fe.pl.BatchCompressor(T, accuracy=0.02, uniform=False, compressor_cfg=dict(orc.DEFAULT_CFG))
    bn.compress(frames, ground=gms)
    buf = bn._buf
    for b, r in enumerate(rows):
        s, tag = r["sha"], (geom, r["frame"], "non-uniform")
        assert sha(buf.key_point_map[b].cpu().numpy()) == s["key_point_map"], tag
        assert sha(buf.salience[b, :r["labels"]].cpu().numpy()) == s["salience"], tag
        n = int(buf.nnz[b])
        assert sha(buf.q16[b, :n].cpu().numpy()) == s["q_nonuniform"], tag


def test_submits_in_flight_own_their_buffers(fe):
    """submit() / collect(): several batches of the SAME size submitted before the first collect() -- single geometry and mixed -- return
    what one compress() per batch returns (every batch in flight owns its output buffers); one submit() more than SLOTS raises."""
    from oracle import oracle as orc
    from rpcc_amd import synth
    names = ["VelodyneVLP16", "Velodyne32E", "VelodyneVLP16", "Velodyne32E"]
    T = {n: fe.ds.build_dataset(lidar_type=n).PCTransformer for n in set(names)}
    def batch(k):
        return [synth.make_frame(4100 + 10 * k + i, orc.GEOMS[n]["H"], orc.GEOMS[n]["W"], vmax_deg=orc.GEOMS[n]["vmax_deg"],
                                 vmin_deg=orc.GEOMS[n]["vmin_deg"]).numpy() for i, n in enumerate(names)]
    batches = [batch(k) for k in range(3)]
    kw = dict(accuracy=0.02, uniform=False, model_method="plane", seed=3)
    mc = fe.pl.MixedBatchCompressor(T, **kw)
    want = [mc.compress(b, names) for b in batches]
    assert want[0] != want[1]
    ctxs = [mc.submit(b, names) for b in batches]            # nothing collected in between
    assert [mc.collect(c) for c in ctxs] == want
    bc = fe.pl.BatchCompressor(T["VelodyneVLP16"], accuracy=0.02, seed=4)
    vlp = [[b[0], b[2]] for b in batches]
    want1 = [bc.compress(v) for v in vlp]
    ctxs = [bc.submit(v) for v in vlp]
    assert [bc.collect(c) for c in reversed(ctxs)][::-1] == want1
    ctxs = [bc.submit(vlp[0]) for _ in range(bc.SLOTS)]
    with pytest.raises(RuntimeError, match="not collected"):
        bc.submit(vlp[0])
    assert all(bc.collect(c) == want1[0] for c in ctxs)
    assert bc.compress(vlp[1]) == want1[1]


def test_entropy_coding_on_a_thread_pool(fe):
    """collect(pool=...) -- the frames' entropy coding on executor threads, as the reference's --workers pool does -- gives the
    same .rpcc strings in the same order."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as orc
    from rpcc_amd import synth
    gd = orc.GEOMS["VelodyneVLP16"]
    T = fe.ds.build_dataset(lidar_type="VelodyneVLP16").PCTransformer
    frames = [synth.make_frame(6000 + i, gd["H"], gd["W"], vmax_deg=gd["vmax_deg"], vmin_deg=gd["vmin_deg"]).numpy() for i in range(9)]
    bc = fe.pl.BatchCompressor(T, accuracy=0.02, seed=2)
    serial = bc.compress(frames)
    with ThreadPoolExecutor(4) as pool:
        assert bc.compress(frames, pool=pool) == serial


def test_streaming_loader_matches_batch_compressor(fe):
    """f4: loader.StreamingCompressor (pinned staging ring, H2D on a copy stream, submit(n+1) before collect(n), entropy
    coding on the pool) gives the .rpcc bytes of BatchCompressor.compress -- full batches, a short last batch, a frame
    without points, .bin-style [N,4] input rows, both frameworks."""
    from oracle import oracle as orc
    from rpcc_amd import synth
    from rpcc_amd.loader import StreamingCompressor
    gd = orc.GEOMS["VelodyneVLP16"]
    ds = fe.ds.build_dataset(lidar_type="VelodyneVLP16")
    frames = [synth.make_frame(700 + i, gd["H"], gd["W"], vmax_deg=gd["vmax_deg"], vmin_deg=gd["vmin_deg"]).numpy() for i in range(11)]
    frames[4] = np.zeros((0, 3), np.float32)
    frames4 = [np.concatenate([f, np.ones((f.shape[0], 1), np.float32)], 1) for f in frames]      # x, y, z, intensity
    ids = [9000 + 7 * i for i in range(len(frames))]
    for kw in (dict(), dict(uniform=False, model_method="plane")):
        bc = fe.pl.BatchCompressor(ds.PCTransformer, accuracy=0.02, seed=5, **kw)
        want = []
        for s in range(0, len(frames), 4):
            want += bc.compress(frames[s:s + 4] + [np.zeros((0, 3), np.float32)] * (4 - len(frames[s:s + 4])),
                                frame_ids=(ids[s:s + 4] + [0] * 4)[:4])[:len(frames[s:s + 4])]
        sc = StreamingCompressor(bc, batch=4, depth=3, workers=4)
        got = {}
        n = sc.run(((frames4[s:s + 4], ids[s:s + 4]) for s in range(0, len(frames), 4)), sink=lambda k, r: got.__setitem__(k, r))
        assert n == len(frames)
        flat = [b for k in sorted(got) for b in got[k]]
        assert flat == want, kw
        raw = {}
        sc.run(((frames[s:s + 4], ids[s:s + 4]) for s in range(0, len(frames), 4)), entropy=False,
               sink=lambda k, r: raw.__setitem__(k, [{kk: np.array(v) for kk, v in r.frame(b).items()} for b in range(len(r))]))
        assert sum(len(v) for v in raw.values()) == len(frames)
        bz = fe.cu.BasicCompressor(method_name="bzip2")
        assert [fe.cu.pack_bitstream(bz.compress_dict(od), uniform=bc.uniform) for k in sorted(raw) for od in raw[k]] == want

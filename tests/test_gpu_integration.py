"""GPU: integration/rpcc_hip_binding.py -- the stub INTEGRATION.md tells a reference maintainer to add -- driven
exactly like the reference's tools/compress.py:93-125 drives its pybind11 modules, against the golden vectors
(outputs of the genuine reference, tests/golden/gen_golden.py).  Only ctypes + the C ABI: nothing of the
rpcc_amd package is imported by the binding."""
import importlib.util
import json
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
MAN = json.load(open(os.path.join(HERE, "golden", "manifest.json")))


@pytest.fixture(scope="module")
def rb():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the -m gpu tests need an MI355X")
    spec = importlib.util.spec_from_file_location("rpcc_hip_binding", os.path.join(ROOT, "integration", "rpcc_hip_binding.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _beq(a, b):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("case", sorted(MAN["cases"]))
def test_binding_reproduces_reference_frame(rb, case):
    from oracle import oracle as orc          # the checker
    c = MAN["cases"][case]
    z = np.load(os.path.join(HERE, "golden", case + ".npz"))
    g = orc.LidarGeom(**orc.GEOMS[c["geom"]])
    tm = orc.transform_map(g)
    # tools/compress.py:96  point_cloud_to_range_image
    ri = rb.point_cloud_to_range_image_even(z["xyz"], g.H, g.W, g.horizontal_FOV, g.vertical_max, g.vertical_min)
    ri_o = orc.project(z["xyz"], g)
    assert _beq(ri[..., 0], ri_o)
    # :100 segment (ground model injected, as the golden generator did)
    seg_idx, centers = rb.segment_range_image(ri[..., 0], tm, z["ground_model"], 100, 0.1)
    assert np.array_equal(seg_idx.astype(np.uint8), z["seg_idx"])
    # :101-102 point model + model_param assembly
    pm = rb.point_modeling(ri, seg_idx)
    nrow = z["model_param"].shape[0]
    assert pm.shape[0] == nrow and _beq(pm[2:], z["model_param"][2:, 3].astype(np.float32))
    # :104-106 prediction + residual
    pred = rb.intra_predict(seg_idx, z["model_param"], tm)
    assert _beq(pred, orc.intra_predict(seg_idx.astype(np.int32), z["model_param"].astype(np.float32), tm))
    residual = ri - pred
    # compress_utils.py:57-81 both quantisers
    q = rb.uniform_quantize(seg_idx, residual, 0.04)
    assert np.array_equal(q.astype(np.int16), z["q_uniform"])
    feat, kp = rb.extract_features_with_segment(ri, seg_idx, 3, 8, 4, 8, 6)
    assert np.array_equal(kp.astype(np.uint8), z["key_point_map"])
    lacc = (np.array([0.04] * 4) + np.array([0, 0.02, 0.04, 0.06])).astype(np.float32)
    qn, sal = rb.nonuniform_quantize(seg_idx, residual, kp, np.array([30, 10, 3, 0], np.int32), lacc, 2)
    assert np.array_equal(qn.astype(np.int16), z["q_nonuniform"])
    assert np.array_equal(sal.astype(np.uint8), z["salience"])
    # compress_utils.py:156 / :206 contour codec round trip
    cm, seq = rb.extract_contour(seg_idx)
    cm_o, seq_o = orc.extract_contour(seg_idx.astype(np.int32))
    assert np.array_equal(cm, cm_o) and np.array_equal(seq, seq_o)
    assert np.array_equal(rb.recover_map(cm, seq), seg_idx)


def test_binding_fps_operator(rb):
    import torch
    from oracle import oracle as orc
    rng = np.random.default_rng(5)
    xyz = rng.normal(0, 10, (3, 3000, 3)).astype(np.float32)
    idx = rb.furthest_point_sample(torch.from_numpy(xyz).cuda(), 64).cpu().numpy()
    for b in range(3):
        assert np.array_equal(idx[b], orc.fps(xyz[b], 64))


def test_binding_error_reporting(rb):
    """Return code + rpcc_last_error() instead of the reference's exit(-1) (ops/fps/src/sampling.cpp:9-21)."""
    import torch
    with pytest.raises(RuntimeError, match="bad argument"):
        rb.furthest_point_sample(torch.zeros((1, 0, 3), dtype=torch.float32, device="cuda"), 4)


def test_binding_and_mirror_classes_at_300_clusters(rb):
    """cluster_num is a free value of the reference's YAML (cfgs/compressor.yaml:22); above 254 a label needs 16 bits.  The STAGE seams the reference
    calls one by one -- segmentation, point model, prediction, the uniform quantiser, the contour codec -- exist in uint16 form up to 1022 clusters
    (rpcc_assign_wide, rpcc_point_model_wide, rpcc_intra_predict_wide, rpcc_predict_quantize_wide, rpcc_contour_*_wide): through the reference-side
    stub and through the mirror classes (PointCloudSegment, QuantizationModule) a VLP-16 sweep at cluster_num = 300 gives the oracle's labels, model
    rows (point and plane), prediction, key points, salience levels and integers of both frameworks; above 1022 the stage seams name their limit."""
    from oracle import oracle as orc
    import rpcc_amd  # noqa: F401
    from rpcc_amd import synth
    from rpcc_amd.segment_utils import PointCloudSegment
    from rpcc_amd.compress_utils import QuantizationModule
    gd = orc.GEOMS["VelodyneVLP16"]
    g = orc.LidarGeom(**gd)
    tm = orc.transform_map(g)
    xyz = synth.make_frame(4242, g.H, g.W, vmax_deg=gd["vmax_deg"], vmin_deg=gd["vmin_deg"]).numpy()
    M = 300
    cfg = dict(orc.DEFAULT_CFG, cluster_num=M)
    gm = np.array([0.01, -0.02, -0.9997, -1.72])
    o = orc.compress_frame(xyz, g, tm, gm, cfg)
    assert int(o["seg_idx"].max()) > 255
    # the reference-side stub
    ri = rb.point_cloud_to_range_image_even(xyz, g.H, g.W, g.horizontal_FOV, g.vertical_max, g.vertical_min)
    seg_idx, _ = rb.segment_range_image(ri[..., 0], tm, gm, M, 0.1)
    assert np.array_equal(seg_idx, o["seg_idx"])
    pm = rb.point_modeling(ri, seg_idx)
    mp = np.asarray(o["model_param"])
    assert pm.shape[0] == mp.shape[0] and _beq(pm[2:], mp[2:, 3].astype(np.float32))
    pred = rb.intra_predict(seg_idx, mp, tm)
    assert _beq(pred, o["pred"])
    q = rb.uniform_quantize(seg_idx, ri - pred, 0.04)
    assert np.array_equal(q, o["q"])
    cm, seq = rb.extract_contour(seg_idx)
    cm_o, seq_o = orc.extract_contour(seg_idx.astype(np.int32))
    assert np.array_equal(cm, cm_o) and np.array_equal(seq, seq_o) and np.array_equal(rb.recover_map(cm, seq), seg_idx)
    o_n = orc.compress_frame(xyz, g, tm, gm, cfg, uniform=False)
    feat, kp = rb.extract_features_with_segment(ri, seg_idx, 3, 8, 4, 8, 6)
    assert np.array_equal(kp.astype(np.uint8), o_n["key_point_map"].astype(np.uint8))
    lacc = (np.array([0.04] * 4) + np.array([0, 0.02, 0.04, 0.06])).astype(np.float32)
    qn_s, sal_s = rb.nonuniform_quantize(seg_idx, ri - pred, kp, np.array([30, 10, 3, 0], np.int32), lacc, 2)
    assert np.array_equal(qn_s, o_n["q"]) and np.array_equal(sal_s, o_n["salience"])
    with pytest.raises(ValueError, match="<= 1022"):
        rb.segment_range_image(ri[..., 0], tm, gm, 1500, 0.1)
    # the mirror classes, driven like tools/compress.py:93-125
    PointCloudSegment.ransac_plane_segmentation = staticmethod(lambda pts, *a, **k: (None, gm))
    try:
        ps = PointCloudSegment(tm)
        pc = orc.backproject(ri[..., 0], tm)
        seg2, gm2 = ps.segment(pc, ri, dict(segment_method="FPS", ground_vertical_threshold=0.1, cluster_num=M))
        assert np.array_equal(seg2, o["seg_idx"]) and np.array_equal(gm2, gm)
        cmod = ps.cluster_modeling(pc, ri, seg2, dict(model_method="point"))
        model_param = np.concatenate((gm.reshape(1, 4), cmod), 0)
        assert _beq(model_param.astype(np.float32), mp.astype(np.float32))
        pred2 = ps.intra_predict(seg2, model_param)
        assert _beq(pred2, o["pred"])
        qm = QuantizationModule(0.04, uniform=True)
        q2, _, _ = qm.quantize_residual(ri - pred2, seg2, pc, ri)
        assert np.array_equal(q2, o["q"])
        from rpcc_amd import _lib
        with pytest.raises(_lib.RpccError, match="cluster_num = 1500.*<= 1022"):
            ps.segment(pc, ri, dict(segment_method="FPS", ground_vertical_threshold=0.1, cluster_num=1500))
        # the plane model stage by stage at 300 clusters (rpcc_plane_model_wide): the oracle's rows for the same seeds
        ps2 = PointCloudSegment(tm, seed=5, frame_id=9)
        cpl = ps2.cluster_modeling(pc, ri, seg2, dict(model_method="plane", angle_threshold=75))
        want = orc.cluster_modeling_plane(pc, ri[..., 0], seg2, tm, 75, 5, 9)
        assert _beq(cpl.astype(np.float32), np.asarray(want).astype(np.float32))
        qn = QuantizationModule(0.04, uniform=False)
        q3, sal3, kp3 = qn.quantize_residual(ri - pred2, seg2, pc, ri)      # key points + salience levels + per-label steps, stage by stage
        assert np.array_equal(q3, o_n["q"]) and np.array_equal(sal3, o_n["salience"]) and np.array_equal(kp3.astype(np.uint8), o_n["key_point_map"].astype(np.uint8))
    finally:
        PointCloudSegment.ransac_plane_segmentation = None

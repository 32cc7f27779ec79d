"""The C restatement (oracle/rpcc_oracle.c) against the reference's own C++ compiled from its
sources into oracle/_ref (`make -C oracle ref`), on seeded random inputs incl. edge cases, plus the
known-answer test of the restated glibc atan2f against this machine's libm.

oracle/_ref is built in the build container (needs /root/reference) and travels to the GPU box as a
built artefact; when it is absent these tests skip (the golden-vector tests still pin the oracle)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from oracle import oracle as orc

REFDIR = os.path.join(os.path.dirname(os.path.abspath(orc.__file__)), "_ref")
have_ref = os.path.exists(os.path.join(REFDIR, ".built"))
needs_ref = pytest.mark.skipif(not have_ref, reason="oracle/_ref not built (needs /root/reference)")


@pytest.fixture(scope="module")
def ref():
    sys.path.insert(0, REFDIR)
    import dataset_utils_cpp, segment_utils_cpp, quantization_utils_cpp, feature_extractor_cpp, contour_utils_cpp  # noqa
    return dict(ds=dataset_utils_cpp, seg=segment_utils_cpp, q=quantization_utils_cpp,
                feat=feature_extractor_cpp, cont=contour_utils_cpp)


def _atan2_inputs(rng, n):
    a = rng.uniform(-80, 80, n).astype(np.float32)
    b = rng.uniform(-80, 80, n).astype(np.float32)
    # elevation-like pairs, random bit patterns (inf/nan/denormals), axis cases
    z = rng.uniform(-30, 5, n).astype(np.float32)
    r = np.abs(rng.uniform(0.5, 80, n)).astype(np.float32)
    bits = rng.integers(0, 2**32, 2 * n, dtype=np.uint64).astype(np.uint32).view(np.float32)
    sp = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 3e38, 2**25, 2**-29], np.float32)
    sy, sx = np.meshgrid(sp, sp)
    y = np.concatenate([a, z, bits[:n], sy.ravel()])
    x = np.concatenate([b, r, bits[n:], sx.ravel()])
    return y, x


def test_atan2f_restatement_matches_libm():
    """KAT: fdlibm restatement == this container's glibc atan2f, bit for bit (NaNs compare as NaN)."""
    rng = np.random.default_rng(7)
    total = 0
    for _ in range(6):
        y, x = _atan2_inputs(rng, 1_000_000)
        o1 = np.empty_like(y)
        o2 = np.empty_like(y)
        orc.lib().orc_atan2f_array(orc._p(y), orc._p(x), orc._p(o1), C.c_long(y.size))
        orc.lib().orc_libm_atan2f_array(orc._p(y), orc._p(x), orc._p(o2), C.c_long(y.size))
        nan = np.isnan(o1) & np.isnan(o2)
        assert np.array_equal(o1.view(np.uint32)[~nan], o2.view(np.uint32)[~nan])
        total += y.size
    assert total >= 18_000_000


def _cloud(rng, n):
    xyz = rng.normal(0, 20, (n, 3)).astype(np.float32)
    xyz[:, 2] = rng.normal(-1, 1.5, n)
    xyz[: n // 50] = xyz[n // 50: 2 * (n // 50)] * np.float32(1.0000001)   # near-duplicates -> pixel collisions
    return xyz


@needs_ref
@pytest.mark.parametrize("geom", sorted(orc.GEOMS))
def test_project_matches_reference_cpp(ref, geom):
    g = orc.LidarGeom(**orc.GEOMS[geom])
    rng = np.random.default_rng(11)
    xyz = _cloud(rng, 150_000)
    xyz[1000] = 0          # depth-0 point resets its pixel (cpp_modules.cpp:459) -- order dependent
    xyz[5] = [1e-30, 2e-30, 0]
    got = orc.project(xyz, g)
    exp = ref["ds"].point_cloud_to_range_image_even(xyz, g.H, g.W, g.horizontal_FOV, g.vertical_max, g.vertical_min)
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))


def _seg_and_ri(rng, h, w, nlab=102, empty_label=None):
    seg = rng.integers(0, nlab, (h, w)).astype(np.int32)
    # spatially coherent runs like real label maps
    seg = np.repeat(seg[:, ::8], 8, axis=1)[:, :w].copy()
    ri = rng.uniform(0.5, 80, (h, w)).astype(np.float32)
    ri[seg == 1] = 0
    if empty_label is not None:
        seg[seg == empty_label] = empty_label + 1
    return seg, ri


@needs_ref
def test_point_modeling_and_predict_match(ref):
    rng = np.random.default_rng(3)
    seg, ri = _seg_and_ri(rng, 64, 2000, empty_label=57)
    pm_o = orc.point_modeling(ri, seg)
    pm_r = ref["seg"].point_modeling(ri.reshape(64, 2000, 1), seg)
    assert np.array_equal(pm_o.view(np.uint32), pm_r.view(np.uint32))
    assert pm_o.view(np.uint32)[57] == 0xFFC00000          # empty label -> NaN with this bit pattern
    tm = orc.transform_map(orc.LidarGeom(**orc.GEOMS["Velodyne64E"]))
    mp = np.concatenate((np.zeros((pm_o.shape[0], 3)), pm_o[:, None].astype(np.float64)), -1)
    mp[0] = [0.0072, -0.054, -0.998, -1.76]
    mp[5] = [0.5, -0.5, 0.1, -3.0]
    mp[6] = [0.5, -0.5, 0.0, 7.0]                          # a+b+c == 0 -> constant prediction
    pr_o = orc.intra_predict(seg, mp, tm)
    pr_r = ref["seg"].intra_predict(seg, mp, tm)
    assert np.array_equal(pr_o.view(np.uint32), pr_r.view(np.uint32))


@needs_ref
def test_quantizers_match(ref):
    rng = np.random.default_rng(4)
    seg, ri = _seg_and_ri(rng, 32, 2250)
    res = rng.normal(0, 1.0, seg.shape).astype(np.float32)
    res.reshape(-1)[:8] = [0.02, 0.06, -0.02, -0.06, 0.1, -0.1, 0.0, 1000.0]   # half-way cases
    q_o = orc.uniform_quantize(seg, res, 0.04)
    q_r = ref["q"].uniform_quantize(seg, res, 0.04)
    assert np.array_equal(q_o, q_r)
    kp = (rng.random(seg.shape) < 0.01).astype(np.int32) * rng.integers(1, 4, seg.shape).astype(np.int32)
    lk, la = np.array([30, 10, 3, 0]), np.array([0.04, 0.06, 0.08, 0.10])
    qn_o, s_o = orc.nonuniform_quantize(seg, res, kp, lk, la, 2)
    qn_r, s_r = ref["q"].nonuniform_quantize(seg, res, kp, lk, la, 2)
    assert np.array_equal(qn_o, qn_r) and np.array_equal(s_o, s_r)


@needs_ref
def test_features_match_on_written_cells(ref):
    """The reference leaves feat/key_point_map uninitialised where it writes nothing
    (cpp_modules.cpp:38-43); compare on the cells the oracle writes and require the reference's
    written key points to be a superset-free match there."""
    rng = np.random.default_rng(5)
    seg, ri = _seg_and_ri(rng, 64, 2000)
    ri = (20 + 5 * np.sin(np.arange(2000) / 40.0)[None, :] + rng.normal(0, 0.05, (64, 2000))).astype(np.float32)
    ri[:, ::97] += 3.0                                                   # gaps -> occlusion gate
    ri[seg == 1] = 0
    f_o, k_o = orc.extract_features_with_segment(ri, seg)
    # make the reference's malloc'd outputs land on zeroed pages: large fresh allocations are mmap'd
    f_r, k_r = ref["feat"].extract_features_with_segment(ri, seg, 3, 8, 4, 8, 6)
    wrote = f_o != 0
    assert np.array_equal(f_o[wrote].view(np.uint32), f_r[wrote].view(np.uint32))
    assert np.array_equal(k_o[k_o > 0], k_r[k_o > 0])
    assert k_o.max() == 3 and (k_o == 1).any() and (k_o == 2).any()


FEATURE_PARAM_DRAWS = [(3, 8, 4, 8, 6)] + [tuple(int(v) for v in r) for r in np.stack([
    np.random.default_rng(77).integers(1, 6, 14), np.random.default_rng(78).integers(2, 13, 14),
    np.random.default_rng(79).integers(0, 7, 14), np.random.default_rng(80).integers(0, 13, 14),
    np.random.default_rng(81).integers(0, 11, 14)], 1)]


_FEAT_CHILD = """
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
import feature_extractor_cpp as fe
z = np.load(sys.argv[2])
p = [int(v) for v in sys.argv[4:9]]
f, k = fe.extract_features_with_segment(z['ri'], z['seg'], *p)      # first call of a fresh process
np.savez(sys.argv[3], f=f, k=k)
"""


@needs_ref
@pytest.mark.parametrize("params", FEATURE_PARAM_DRAWS)
def test_features_parameter_sweep_matches(params, tmp_path):
    """Key-point extraction with non-default feature_region / segments / sharp / less_sharp / flat counts (incl. zeros and
    less_sharp < sharp): the restatement against the reference's C++.  The reference leaves unwritten cells uninitialised
    (cpp_modules.cpp:38-43), so it runs once in a fresh process with outputs large enough (288 KB) to be mmap'd zero pages:
    then the whole key point map can be compared, not only the cells the restatement writes."""
    import subprocess
    fr, segments, sharp, less, flat = params
    rng = np.random.default_rng(1000 + 7 * fr + segments)
    h, w = 48, 1500
    seg = np.repeat(rng.integers(0, 30, (h, (w + 5) // 6)), 6, axis=1)[:, :w].astype(np.int32)
    ri = (20 + 5 * np.sin(np.arange(w) / 23.0)[None, :] + rng.normal(0, 0.04, (h, w))).astype(np.float32)
    ri[:, ::61] += 2.0
    ri[3, :] = 17.0                                                      # constant row: zero curvatures, ties
    seg[5, 40:] = 1                                                      # too few valid pixels
    ri[seg == 1] = 0
    f_o, k_o = orc.extract_features_with_segment(ri, seg, fr, segments, sharp, less, flat)
    np.savez(tmp_path / "in.npz", ri=ri, seg=seg)
    subprocess.check_call([sys.executable, "-c", _FEAT_CHILD, REFDIR, str(tmp_path / "in.npz"), str(tmp_path / "out.npz")]
                          + [str(v) for v in params])
    z = np.load(tmp_path / "out.npz")
    f_r, k_r = z["f"], z["k"]
    wrote = f_o != 0
    assert np.array_equal(f_o[wrote].view(np.uint32), f_r[wrote].view(np.uint32)), params
    assert np.array_equal(k_o, k_r), params


@needs_ref
def test_contour_roundtrip_matches(ref):
    rng = np.random.default_rng(6)
    seg, _ = _seg_and_ri(rng, 16, 1800)
    cm_o, sq_o = orc.extract_contour(seg)
    cm_r, sq_r = ref["cont"].extract_contour(seg)
    assert np.array_equal(cm_o, cm_r) and np.array_equal(sq_o, sq_r)
    assert np.array_equal(orc.recover_map(cm_o, sq_o), ref["cont"].recover_map(cm_r, sq_r))
    assert np.array_equal(orc.recover_map(cm_o, sq_o), seg)


def test_numpy_rows_match_c_rows():
    """a5/a7 written with the reference's NumPy expressions == the C rows (small case)."""
    g = orc.LidarGeom(H=16, W=200, vmax_deg=15, vmin_deg=-15)
    tm = orc.transform_map(g)
    rng = np.random.default_rng(8)
    ri = rng.uniform(1, 60, (16, 200)).astype(np.float32)
    ri[rng.random(ri.shape) < 0.2] = 0
    pc = orc.backproject(ri, tm)
    plane = np.array([0.01, -0.02, -0.999, -1.7])
    assert np.array_equal(orc.vertical_residual(pc, plane), orc.np_vertical_residual(pc, plane))
    cen = pc.reshape(-1, 3)[rng.choice(3200, 100, replace=False)]
    assert np.array_equal(orc.assign(ri, pc, tm, plane, cen), orc.np_assign(ri, pc, tm, plane, cen))


def test_np_mean_f32_restatement():
    rng = np.random.default_rng(9)
    for n in [1, 2, 7, 8, 9, 29, 127, 128, 129, 1000, 8191, 8192, 8193, 20000, 128000]:
        a = rng.uniform(0.5, 80, n).astype(np.float32)
        assert orc.np_mean_f32(a).view(np.uint32) == a.reshape(n, 1).mean().view(np.uint32), n

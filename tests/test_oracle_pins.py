"""The CPU oracle against the round-2 fixture set (tests/golden/pins_*.npz, produced by gen_golden_pins.py from the GENUINE
reference): cluster_modeling('plane') + plane_angle_validation with injected plane rows (a9 glue), everything downstream
of it (prediction with plane rows, both quantisers, the .rpcc bytes), and the decoder (f3)."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
PINS = json.load(open(os.path.join(HERE, "golden", "pins_manifest.json")))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load_case(name):
    c = PINS[name]
    base = np.load(os.path.join(HERE, "golden", c["base"] + ".npz"))
    z = np.load(os.path.join(HERE, "golden", "pins_" + name + ".npz"))
    xyz = base["xyz"]
    if c["min_range"] > 0:
        xyz = np.ascontiguousarray(xyz[np.sqrt((xyz.astype(np.float64) ** 2).sum(1)) >= c["min_range"]])
    assert xyz.shape[0] == c["n_points"]
    g = orc.LidarGeom(**orc.GEOMS[c["geom"]])
    return c, z, xyz, base["ground_model"], g, orc.transform_map(g), base


@pytest.fixture(scope="module", params=sorted(PINS))
def case(request):
    return (request.param,) + load_case(request.param)


def test_fixture_covers_the_branches():
    """The injected rows exercise acceptance, rejection by the angle check, NaN rows, small labels and an empty label."""
    assert any(c["empty_labels"] > 0 for c in PINS.values())
    assert all(c["small_labels"] > 0 for c in PINS.values())
    assert any(c["nan_rows"] > 0 for c in PINS.values())
    assert all(0 < c["plane_rows_kept"] < c["ransac_calls"] for c in PINS.values())


def test_plane_model_glue_matches_reference(case):
    name, c, z, xyz, gm, g, tm, _ = case
    ri = orc.project(xyz, g)
    s = orc.segment(ri, tm, gm)
    with np.errstate(all="ignore"):
        cm = orc.cluster_modeling_plane(s["pc"], ri, s["seg_idx"], tm, 75, inject=list(z["plane_rows"]))
    assert cm.shape == z["cluster_models"].shape
    assert np.array_equal(cm.view(np.uint64), z["cluster_models"].view(np.uint64))
    assert sha(cm) == c["sha"]["cluster_models"]
    mp = np.concatenate((np.asarray(gm, np.float64).reshape(1, 4), cm), 0)
    for uniform, qk, rk in ((True, "q_uniform_plane", "rpcc_uniform_plane"), (False, "q_nonuniform_plane", "rpcc_nonuniform_plane")):
        o = orc.compress_frame(xyz, g, tm, gm, uniform=uniform, model_param=mp)
        if uniform:
            assert sha(o["pred"]) == c["sha"]["pred_plane"]
        else:
            assert np.array_equal(o["salience"].astype(np.uint8), z["salience_plane"])
        assert np.array_equal(o["q"].astype(np.int16), z[qk])
        blob = orc.bitstream_bytes(orc.pack_payload(mp, o["seg_idx"], o["salience"], o["q"]), uniform=uniform)
        assert blob == z[rk].tobytes()
        assert hashlib.sha256(blob).hexdigest() == c["sha"][rk]


@pytest.mark.parametrize("kind", ["uniform_point", "uniform_plane", "nonuniform_plane"])
def test_decoder_matches_reference(case, kind):
    name, c, z, xyz, gm, g, tm, base = case
    if kind == "uniform_point":
        blob = (z["rpcc_uniform_point"] if "rpcc_uniform_point" in z.files else base["rpcc"]).tobytes()
    else:
        blob = z["rpcc_" + kind].tobytes()
    d = orc.decode_frame(blob, g, tm, 0.02, uniform=kind.startswith("uniform"))
    assert sha(d["residual"].astype(np.float32)) == c["sha"]["residual_" + kind]
    assert sha(d["ri_rec"].astype(np.float32)) == c["sha"]["ri_rec_" + kind]
    assert sha(d["pc_rec"]) == c["sha"]["pc_rec_" + kind]
    if "ri_rec_" + kind in z.files:
        assert np.array_equal(d["ri_rec"].astype(np.float32).view(np.uint32), z["ri_rec_" + kind].view(np.uint32))
    ri = orc.project(xyz, g)
    err = float(np.abs(d["ri_rec"][..., 0] - ri)[ri != 0].max())
    exp = c["max_err"][kind]
    assert (np.isnan(err) and np.isnan(exp)) or abs(err - exp) < 1e-12     # NaN: labels modelled by an injected NaN row
    if kind == "uniform_point":
        assert err <= 0.02 + 1e-5                                           # README.md:101-106


def test_fps_modes_closed_form_equals_the_cuda_kernel_thread_by_thread():
    """a6, the CUDA binary's unpinned degrees of freedom (oracle.fps_modes): the closed form of the reduction tree's tie rule
    -- smallest bit-reversed (k mod block), then smallest k -- against the kernel restated thread by thread (strided scan per
    thread with a strict '>', the shared-memory tree of `__update`, sampling_gpu.cu:16-21,55-69,74-134), on inputs made of
    ties (integer lattices, duplicates), for the un-fused distance and both contractions; mode (0, lowest index) is orc.fps."""
    from oracle import oracle as orc
    rng = np.random.default_rng(7)
    differs = 0
    for trial in range(30):
        n, m = int(rng.integers(3, 2500)), int(rng.integers(1, 40))
        pts = rng.integers(-3, 4, (n, 3)).astype(np.float32)
        for f in (0, 1, 2):
            a = orc.fps_modes(pts, m, f, True)
            assert np.array_equal(a, orc.fps_cuda_emulated(pts, m, f)), (trial, f)
            differs += int(not np.array_equal(a, orc.fps_modes(pts, m, f, False)))
        assert np.array_equal(orc.fps_modes(pts, m, 0, False), orc.fps(pts, m))
    assert differs > 30          # the tie rule matters on these inputs
    # opt_n_threads (sampling_gpu.cu:9-13)
    assert [orc.lib().orc_fps_block_size(n) for n in (1, 2, 3, 7, 9, 1023, 1025, 5000, 100000)] == [1, 2, 2, 4, 8, 512, 1024, 1024, 1024]
    # a contraction changes roundings: on generic points the three distance forms must at least run and agree on the first centres
    pts = rng.normal(0, 10, (4000, 3)).astype(np.float32)
    i0, i1, i2 = (orc.fps_modes(pts, 50, f, False) for f in (0, 1, 2))
    assert i0[0] == i1[0] == i2[0] == 0 and i0[1] == i1[1] == i2[1]


def tie_fixture():
    """tests/golden/pins_ties_vlp16.npz (gen_golden_ties.py: the reference's own segment() on a constructed image and a prescribed centre
    list) -> (geometry, ray table, range image, ground model, centres f32 [100,3], labels the reference produced)."""
    z = np.load(os.path.join(HERE, "golden", "pins_ties_vlp16.npz"))
    man = json.load(open(os.path.join(HERE, "golden", "pins_ties_manifest.json")))
    g = orc.LidarGeom(**orc.GEOMS[man["geom"]])
    tm = orc.transform_map(g)
    ri = np.repeat(z["row_ranges"][:, None], g.W, 1).astype(np.float32)
    ri.reshape(-1)[::int(z["empty_stride"])] = 0
    ri.reshape(-1)[z["tuned_pixels"]] = z["tuned_ranges"]
    pc = orc.backproject(ri, tm)
    cen = pc.reshape(-1, 3)[z["centre_pixels"]].astype(np.float32)
    return man, g, tm, ri, z["ground_model"], cen, z["seg_idx"]


def test_assignment_tie_order_matches_reference():
    """a7: np.argmax(-np.abs(distance)) keeps the FIRST maximum -- the ground before a cluster, the lower cluster index among fp32 radii that are
    equal although their squared distances differ, and among exact duplicates (utils/segment_utils.py:21-23,127-131).  The labels come from
    the genuine reference; 114 of the pixels carry a label that is NOT the arg-min of the squared distance."""
    man, g, tm, ri, gm, cen, want = tie_fixture()
    assert man["decided_by_tie_order"] >= 100 and sha(want) == man["sha_seg_idx"]
    pc = orc.backproject(ri, tm)
    got = orc.assign(ri, pc, tm, gm, cen)
    assert np.array_equal(got.astype(np.uint8), want)
    assert np.array_equal(orc.np_assign(ri.reshape(g.H, g.W, 1), pc, tm, gm, cen).astype(np.uint8), want)
    # the fixture does what it says: for those pixels a squared-distance arg-min would answer differently
    d = pc.reshape(-1, 1, 3) - cen[None]
    d2 = ((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]).astype(np.float32)
    lab = want.reshape(-1).astype(np.int64)
    assert int(((lab >= 2) & (lab - 2 != d2.argmin(1)) & (ri.reshape(-1) != 0)).sum()) >= man["decided_by_tie_order"]

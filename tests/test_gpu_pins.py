"""-m gpu: the HIP path against the round-2 fixtures produced from the GENUINE reference (tests/golden/pins_*.npz):
a9 plane model glue with injected plane rows (angle validation, <30-pixel and empty labels, NaN rows), everything downstream
(prediction with plane rows, both quantisers, .rpcc bytes) and the decoder f3 (rpcc_contour_decode + rpcc_decode)."""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
PINS = json.load(open(os.path.join(HERE, "golden", "pins_manifest.json")))
LIDAR_OF = {"Velodyne64E": "Velodyne64E", "Velodyne64E_2048": "Velodyne64E_2048", "Velodyne32E": "Velodyne32E", "VelodyneVLP16": "VelodyneVLP16"}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available()
    import rpcc_amd  # noqa: F401
    from rpcc_amd import ops, compress_utils, dataset
    from rpcc_amd.tools import decompress
    from oracle import oracle as orc
    return dict(torch=torch, ops=ops, orc=orc, cu=compress_utils, ds=dataset, dec=decompress, dev=torch.device("cuda:0"))


def _case(env, name):
    orc = env["orc"]
    c = PINS[name]
    base = np.load(os.path.join(HERE, "golden", c["base"] + ".npz"))
    z = np.load(os.path.join(HERE, "golden", "pins_" + name + ".npz"))
    xyz = base["xyz"]
    if c["min_range"] > 0:
        xyz = np.ascontiguousarray(xyz[np.sqrt((xyz.astype(np.float64) ** 2).sum(1)) >= c["min_range"]])
    g = orc.LidarGeom(**orc.GEOMS[c["geom"]])
    return c, z, xyz, base, g, orc.transform_map(g)


@pytest.mark.parametrize("name", sorted(PINS))
def test_plane_model_glue_matches_reference(env, name):
    """rpcc_plane_model with the fixture's plane rows injected in place of its RANSAC == the genuine
    cluster_modeling('plane') with the same rows injected through ransac_plane_segmentation; then prediction + quantisers +
    container on top give the reference's .rpcc bytes."""
    torch, ops, orc, dev = env["torch"], env["ops"], env["orc"], env["dev"]
    c, z, xyz, base, g, tm = _case(env, name)
    gm = base["ground_model"]
    geom = ops.make_geom(g.H, g.W, g.horizontal_FOV, g.vertical_max, g.vertical_min)
    d_tm = torch.from_numpy(tm).to(dev)
    offs = torch.tensor([0, xyz.shape[0]], dtype=torch.int64, device=dev)
    ri = ops.project(torch.from_numpy(xyz).to(dev), offs, geom)
    ground = torch.from_numpy(gm.reshape(1, 4)).to(dev)
    temp, info, tab = ops.ground_mask(ri, d_tm, ground, 0.1, fps_table=True)
    _, centers = ops.fps_range(ri, d_tm, temp, info, 100, fps_table=tab)
    seg = ops.assign(ri, d_tm, ground, centers)
    seg_h = seg[0].cpu().numpy()
    counts = np.bincount(seg_h.reshape(-1), minlength=102)
    inj = np.zeros((1, 102, 4), np.float64)
    rows = list(z["plane_rows"])
    for k in range(2, int(seg_h.max()) + 1):
        if counts[k] >= 30:
            inj[0, k] = rows.pop(0)
    assert not rows
    model = ops.plane_model(ri, d_tm, seg, 100, angle_threshold=75, seed=0, ground=ground, inject=torch.from_numpy(inj).to(dev))
    nrow = int(seg_h.max()) + 1
    exp = np.concatenate((gm.reshape(1, 4), z["cluster_models"]), 0).astype(np.float32)
    assert np.array_equal(model[0, :nrow].cpu().numpy().view(np.uint32), exp.view(np.uint32))
    lacc = (np.array([0.04] * 4) + np.array([0, 0.02, 0.04, 0.06])).astype(np.float32)
    for uniform, qk, rk in ((True, "q_uniform_plane", "rpcc_uniform_plane"), (False, "q_nonuniform_plane", "rpcc_nonuniform_plane")):
        label_acc = sal = None
        if not uniform:
            _, kp = ops.extract_features(ri, seg)
            sal, label_acc = ops.salience(seg, kp, (30, 10, 3, 0), lacc, 2, 100)
            assert np.array_equal(sal[0, :nrow].cpu().numpy(), z["salience_plane"])
        q, nnz, pred = ops.predict_quantize(ri, d_tm, seg, model, 0.04, 100, want_pred=True, int16=True, label_acc=label_acc)
        n = int(nnz[0])
        if uniform:
            assert sha(pred[0].cpu().numpy().reshape(g.H, g.W, 1)) == c["sha"]["pred_plane"]
        assert np.array_equal(q[0, :n].cpu().numpy(), z[qk])
        bits, seq, nseq = ops.contour_encode(seg)
        od = {"residual_quantized": q[0, :n].cpu().numpy()}
        if not uniform:
            od["salience_level"] = sal[0, :nrow].cpu().numpy()
        od["contour_map"] = bits[0].cpu().numpy()
        od["idx_sequence"] = seq[0, :int(nseq[0])].cpu().numpy().view(np.uint16)
        od["plane_param"] = model[0, :nrow].cpu().numpy()
        blob = env["cu"].pack_bitstream(env["cu"].BasicCompressor(method_name="bzip2").compress_dict(od), uniform=uniform)
        assert blob == z[rk].tobytes()


@pytest.mark.parametrize("name", sorted(PINS))
@pytest.mark.parametrize("kind", ["uniform_point", "uniform_plane", "nonuniform_plane"])
def test_decoder_matches_reference(env, name, kind):
    """tools.decompress.decode_frame (rpcc_contour_decode + rpcc_decode) on the reference's own bitstreams == the genuine
    decompress_point_cloud + dequantize_residual + intra_predict + range_image_to_point_cloud."""
    c, z, xyz, base, g, tm = _case(env, name)
    uniform = kind.startswith("uniform")
    if kind == "uniform_point":
        blob = (z["rpcc_uniform_point"] if "rpcc_uniform_point" in z.files else base["rpcc"]).tobytes()
    else:
        blob = z["rpcc_" + kind].tobytes()
    T = env["ds"].build_dataset(lidar_type=LIDAR_OF[c["geom"]]).PCTransformer
    lacc = np.array([0.04] * 4) + np.array([0, 0.02, 0.04, 0.06])
    rec, pc, seg = env["dec"].decode_frame(env["cu"].unpack_bitstream(blob, uniform=uniform), env["cu"].BasicCompressor(method_name="bzip2"),
                                           T, 100, 0.04, lacc, uniform)
    assert sha(rec.astype(np.float32).reshape(g.H, g.W, 1)) == c["sha"]["ri_rec_" + kind]
    assert sha(pc.astype(np.float32).reshape(g.H, g.W, 3)) == c["sha"]["pc_rec_" + kind]
    if "ri_rec_" + kind in z.files:
        assert np.array_equal(rec.reshape(-1).view(np.uint32), z["ri_rec_" + kind].reshape(-1).view(np.uint32))


def test_decoder_rejects_inconsistent_streams(env):
    """The .rpcc file stores no configuration: a wrong lidar geometry or cluster count must raise, not index out of bounds."""
    c, z, xyz, base, g, tm = _case(env, "synth_vlp16")
    cu = env["cu"]
    blob = base["rpcc"].tobytes()
    bc = cu.BasicCompressor(method_name="bzip2")
    T16 = env["ds"].build_dataset(lidar_type="VelodyneVLP16").PCTransformer
    T32 = env["ds"].build_dataset(lidar_type="Velodyne32E").PCTransformer
    d = cu.unpack_bitstream(blob, uniform=True)
    env["dec"].decode_frame(d, bc, T16, 100, 0.04, None, True)                       # the right configuration decodes
    with pytest.raises(ValueError):
        env["dec"].decode_frame(d, bc, T32, 100, 0.04, None, True)                   # wrong --lidar
    with pytest.raises(ValueError):
        env["dec"].decode_frame(d, bc, T16, 50, 0.04, None, True)                    # encoded with more clusters than configured
    raw = bc.decompress_dict(d)
    short = dict(raw, residual_quantized=raw["residual_quantized"][:-8])
    with pytest.raises(ValueError):
        env["dec"].decode_frame(bc.compress_dict({k: np.frombuffer(v, np.uint8) for k, v in short.items()}), bc, T16, 100, 0.04, None, True)
    # non-uniform framework: the salience levels come straight from the file and index the step table
    nrow = np.frombuffer(raw["plane_param"], np.float32).size // 4
    lv = [0.04, 0.06, 0.08, 0.10]
    as_u8 = lambda dct: bc.compress_dict({k: np.frombuffer(v, np.uint8) for k, v in dct.items()})
    ok = dict(raw, salience_level=np.full(nrow, 3, np.uint8).tobytes())
    env["dec"].decode_frame(as_u8(ok), bc, T16, 100, 0.04, lv, False)                # a consistent stream decodes
    with pytest.raises(ValueError):                                                  # a level the configuration does not define
        env["dec"].decode_frame(as_u8(dict(raw, salience_level=np.full(nrow, 4, np.uint8).tobytes())), bc, T16, 100, 0.04, lv, False)
    with pytest.raises(ValueError):                                                  # fewer levels than model rows
        env["dec"].decode_frame(as_u8(dict(raw, salience_level=np.zeros(nrow - 1, np.uint8).tobytes())), bc, T16, 100, 0.04, lv, False)
    with pytest.raises(ValueError):                                                  # a uniform stream read as non-uniform
        env["dec"].decode_frame(as_u8(raw), bc, T16, 100, 0.04, lv, False)


def test_assignment_tie_order_matches_reference(env):
    """rpcc_assign on the tie fixture (tests/golden/pins_ties_vlp16.npz: labels produced by the GENUINE reference's segment() on a constructed
    image with a prescribed centre list): fp32 radii that are equal although the squared distances differ, exact duplicate centres, empty
    pixels -- the first maximum of np.argmax(-np.abs(distance)) (utils/segment_utils.py:21-23,127-131,168-169), bit for bit."""
    from test_oracle_pins import tie_fixture
    torch, ops, dev = env["torch"], env["ops"], env["dev"]
    man, g, tm, ri, gm, cen, want = tie_fixture()
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    seg = ops.assign(to(ri[None]), to(tm), to(np.asarray(gm, np.float64).reshape(1, 4)), to(cen[None]))[0].cpu().numpy()
    bad = np.flatnonzero(seg.reshape(-1) != want.reshape(-1))
    assert bad.size == 0, (bad[:8], seg.reshape(-1)[bad[:8]], want.reshape(-1)[bad[:8]])

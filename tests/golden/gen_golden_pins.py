#!/usr/bin/env python3
"""Second fixture set (round 2), again produced by running the GENUINE reference Python here (build container only;
needs /root/reference and oracle/_ref).  It pins the two seams the first set left to the oracle's own restatement:

  a9  cluster_modeling('plane') + plane_angle_validation (utils/segment_utils.py:84-93,188-216): the reference's
      Python is run with model_method='plane' and INJECTED plane rows in place of Open3D's random segment_plane --
      rows that pass the 75 degree check, rows that fail it (grazing planes), non-normalised rows, NaN rows
      (NaN compares false => accepted), next to the labels the frame itself provides below 30 pixels / without pixels.
      Stored: the injected rows (inputs) and cluster_models, pred / q / .rpcc of the uniform and the non-uniform
      framework on top of them (expected outputs).
  f3  the decoder: decompress_point_cloud + QuantizationModule.dequantize_residual + intra_predict + range_image_rec +
      range_image_to_point_cloud (utils/compress_utils.py:114-132,199-214, tools/decompress.py:88-112) run on the
      reference's own bitstreams (uniform + point, uniform + plane, non-uniform + plane).  Stored: the .rpcc bytes
      (inputs) and the reconstructed range image / point cloud (expected outputs; arrays for the small geometry,
      SHA-256 for the large ones).

Inputs are the frames of the first set (tests/golden/<case>.npz: xyz, ground_model).  Nothing of the reference travels."""
import hashlib
import json
import os
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  (stubs, run_reference, write_lidar_yaml)

import numpy as np  # noqa: E402

orc = gg.orc
REF = gg.REF


def make_plane_rows(pc, tm, seg, rng):
    """One injected row per label (>= 2) with at least 30 pixels, in the order cluster_modeling asks for them."""
    rows, kinds = [], []
    for i in range(2, int(seg.max()) + 1):
        idx = np.where(seg == i)
        if idx[0].shape[0] < 30:
            continue
        pts = pc[idx].astype(np.float64)
        c = pts.mean(0)
        _, _, vt = np.linalg.svd(pts - c, full_matrices=False)
        n = vt[-1]
        kind = len(rows) % 8
        if kind in (0, 1, 2, 3):                      # a sensible fit (least squares), unit normal
            row = [n[0], n[1], n[2], -n @ c]
        elif kind == 4:                               # the same plane, not normalised (the check divides by |n|)
            row = list(3.7 * np.array([n[0], n[1], n[2], -n @ c]))
        elif kind == 5:                               # grazing plane: normal perpendicular to the mean ray -> rejected
            ray = tm[idx].astype(np.float64).mean(0)
            ray /= np.linalg.norm(ray)
            m = np.cross(ray, [0.0, 0.0, 1.0])
            m /= np.linalg.norm(m)
            row = [m[0], m[1], m[2], -m @ c]
        elif kind == 6:                               # tilted by about 72 degrees from the mean ray: near the threshold
            ray = tm[idx].astype(np.float64).mean(0)
            ray /= np.linalg.norm(ray)
            m = np.cross(ray, [0.0, 0.0, 1.0])
            m /= np.linalg.norm(m)
            t = np.cos(np.radians(72.0)) * ray + np.sin(np.radians(72.0)) * m
            row = [t[0], t[1], t[2], -t @ c]
        else:                                         # NaN row: arccos(NaN).max() > thr is False -> accepted as it is
            row = [np.nan, np.nan, np.nan, np.nan] if rng.random() < 0.5 else [n[0], n[1], n[2], -n @ c]
        rows.append(row)
        kinds.append(kind)
    return np.asarray(rows, np.float64), kinds


def run_decoder(rpcc, uniform, lidar_yaml, accuracy=0.02):
    """tools/decompress.py:88-112 through the genuine reference modules."""
    from dataset.transformer import PCTransformer
    from utils.segment_utils import PointCloudSegment
    from utils.compress_utils import QuantizationModule, decompress_point_cloud, BasicCompressor
    import yaml
    cfg = yaml.safe_load(open(os.path.join(REF, "cfgs/compressor.yaml")))
    acc = accuracy * 2
    T = PCTransformer(lidar_yaml, None)
    H, W = T.transform_map.shape[:2]
    keys = ([] if uniform else ["salience_level"]) + ["contour_map", "idx_sequence", "plane_param", "residual_quantized"]
    cd, off = {}, 0
    for k in keys:                                   # read_compressed_bitstream (utils/compress_utils.py:181-196) on bytes
        (n,) = struct.unpack("i", rpcc[off:off + 4])
        cd[k] = rpcc[off + 4: off + 4 + n]
        off += 4 + n
    assert off == len(rpcc)
    bc = BasicCompressor(method_name="bzip2")
    model_num = cfg["cluster_num"] + 1               # tools/decompress.py:75 (one short of the rows: SURVEY 8c)
    rq, seg_idx, sal, plane_param = decompress_point_cloud(cd, bc, model_num, H, W)
    if uniform:
        QM = QuantizationModule(acc)
    else:
        QM = QuantizationModule(acc, uniform=False, level_kp_num=tuple(cfg["level_key_point_num"]),
                                level_dacc=tuple(cfg["level_delta_acc"]), ground_salience_level=cfg["ground_salience_level"],
                                feature_region=cfg["feature_region"], segments=cfg["segments"], sharp_num=cfg["sharp_num"],
                                less_sharp_num=cfg["less_sharp_num"], flat_num=cfg["flat_num"])
    residual = QM.dequantize_residual(rq, seg_idx, sal)
    ps = PointCloudSegment(T.transform_map)
    pred = ps.intra_predict(seg_idx, plane_param)
    ri_rec = pred + residual
    pc_rec = T.range_image_to_point_cloud(ri_rec)
    return dict(seg_idx=seg_idx, residual=residual, pred=pred, ri_rec=ri_rec, pc_rec=pc_rec)


def rpcc_of(ref, uniform):
    from utils.compress_utils import compress_point_cloud, BasicCompressor
    bc = BasicCompressor(method_name="bzip2")
    od, cd = compress_point_cloud(bc, ref["model_param"], ref["seg_idx"], ref["salience"], ref["q"], ref["pc"], ref["ri"], full=False)
    parts = []
    for k in ([] if uniform else ["salience_level"]) + ["contour_map", "idx_sequence", "plane_param", "residual_quantized"]:
        parts += [struct.pack("i", len(cd[k])), cd[k]]
    return b"".join(parts)


def main():
    gg.install_stubs()
    tmp = "/tmp/rpcc_golden_tmp"
    os.makedirs(tmp, exist_ok=True)
    man = json.load(open(os.path.join(HERE, "manifest.json")))
    pins = {}
    # (case, base fixture, points nearer than this many metres removed).  The last case empties the label of the centre
    # that sits at the sensor origin (the first FPS candidate is an empty pixel): an empty label -> NaN mean row.
    todo = [(n, n, 0.0) for n in sorted(man["cases"])] + [("synth_vlp16_far", "synth_vlp16", 9.0)]
    for name, base, min_range in todo:
        geom = man["cases"][base]["geom"]
        g = orc.GEOMS[geom]
        z = np.load(os.path.join(HERE, base + ".npz"))
        xyz, gm = z["xyz"], z["ground_model"]
        if min_range > 0:
            xyz = np.ascontiguousarray(xyz[np.sqrt((xyz.astype(np.float64) ** 2).sum(1)) >= min_range])
        yml = os.path.join(tmp, geom + ".yaml")
        gg.write_lidar_yaml(yml, g)
        rng = np.random.default_rng(4242)
        ref_pt = gg.run_reference(xyz, yml, gm, uniform=True)                 # segmentation to size the injected rows
        seg = ref_pt["seg_idx"]
        rows, kinds = make_plane_rows(ref_pt["pc"], ref_pt["tm"], seg, rng)
        counts = np.bincount(seg.reshape(-1), minlength=int(seg.max()) + 1)
        ref_u = gg.run_reference(xyz, yml, gm, uniform=True, plane_rows=[list(r) for r in rows])
        ref_n = gg.run_reference(xyz, yml, gm, uniform=False, plane_rows=[list(r) for r in rows])
        assert np.array_equal(ref_u["seg_idx"], seg) and np.array_equal(ref_n["seg_idx"], seg)
        cm = ref_u["model_param"][1:]
        assert np.array_equal(cm.view(np.uint64), ref_n["model_param"][1:].view(np.uint64))
        # zero-initialised key-point semantics (see gen_golden.py): the genuine C++ quantiser on the oracle's map
        _, kp_o = orc.extract_features_with_segment(ref_n["ri"][..., 0], seg)
        lacc = np.array([0.04] * 4) + np.array([0, 0.02, 0.04, 0.06])
        from ops.cpp_modules import quantization_utils_cpp
        q_n, sal_n = quantization_utils_cpp.nonuniform_quantize(seg, ref_n["residual"], kp_o, np.array([30, 10, 3, 0]), lacc, 2)
        ref_n = dict(ref_n, q=q_n, salience=sal_n)
        rp_u_pt = z["rpcc"].tobytes() if min_range == 0 else ref_pt["rpcc"]
        rp_u_pl = ref_u["rpcc"]
        rp_n_pl = rpcc_of(ref_n, uniform=False)
        dec = {"uniform_point": run_decoder(rp_u_pt, True, yml), "uniform_plane": run_decoder(rp_u_pl, True, yml),
               "nonuniform_plane": run_decoder(rp_n_pl, False, yml)}
        for k, d in dec.items():                      # the decoder recovers the labels; the error bound of the README holds
            assert np.array_equal(d["seg_idx"], seg), (name, k)
        accepted = int(sum(1 for r in cm[1:] if not (r[0] == 0 and r[1] == 0 and r[2] == 0)))
        small = int(((counts[2:] < 30) & (counts[2:] > 0)).sum())
        empty = int((counts[2:] == 0).sum())
        nan_rows = int(np.isnan(cm[:, 3]).sum())
        print(name, "labels", len(counts), "ransac calls", len(rows), "plane rows kept", accepted, "small labels", small,
              "empty labels", empty, "NaN rows", nan_rows)
        store_arrays = geom == "VelodyneVLP16"
        out = dict(plane_rows=rows, plane_row_kinds=np.asarray(kinds, np.int8), cluster_models=cm,
                   q_uniform_plane=ref_u["q"].astype(np.int16), q_nonuniform_plane=q_n.astype(np.int16),
                   salience_plane=sal_n.astype(np.uint8),
                   rpcc_uniform_plane=np.frombuffer(rp_u_pl, np.uint8), rpcc_nonuniform_plane=np.frombuffer(rp_n_pl, np.uint8))
        if store_arrays:
            for k, d in dec.items():
                out["ri_rec_" + k] = d["ri_rec"].astype(np.float32)
        np.savez_compressed(os.path.join(HERE, "pins_" + name + ".npz"), **out)
        if min_range > 0:
            out["rpcc_uniform_point"] = np.frombuffer(rp_u_pt, np.uint8)
            np.savez_compressed(os.path.join(HERE, "pins_" + name + ".npz"), **out)
        pins[name] = dict(geom=geom, base=base, min_range=min_range, n_points=int(xyz.shape[0]), ransac_calls=len(rows), plane_rows_kept=accepted, small_labels=small, empty_labels=empty,
                          nan_rows=nan_rows,
                          sha=dict(cluster_models=gg.sha(cm), pred_plane=gg.sha(ref_u["pred"]),
                                   rpcc_uniform_plane=hashlib.sha256(rp_u_pl).hexdigest(),
                                   rpcc_nonuniform_plane=hashlib.sha256(rp_n_pl).hexdigest(),
                                   **{"ri_rec_" + k: gg.sha(d["ri_rec"].astype(np.float32)) for k, d in dec.items()},
                                   **{"pc_rec_" + k: gg.sha(d["pc_rec"].astype(np.float32)) for k, d in dec.items()},
                                   **{"residual_" + k: gg.sha(d["residual"].astype(np.float32)) for k, d in dec.items()}),
                          max_err=dict((k, float(np.abs(d["ri_rec"] - ref_pt["ri"])[ref_pt["ri"] != 0].max())) for k, d in dec.items()))
    json.dump(pins, open(os.path.join(HERE, "pins_manifest.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()

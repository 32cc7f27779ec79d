#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the GENUINE reference here.

Runs only in the build container (needs /root/reference and oracle/_ref built by
`make -C oracle ref`).  Nothing of the reference travels: this script imports the reference's
Python (utils/segment_utils.py, utils/compress_utils.py, dataset/transformer.py) with in-memory
stubs for the packages the image lacks, feeds it seeded inputs, and stores INPUTS + EXPECTED
OUTPUTS as data (npz / sha256).  The committed fixtures are what tests/test_oracle_golden.py and the
`-m gpu` parity tests check against on any machine.

Stubs / injections:
  * IPython, easydict, lz4, open3d  -> trivial stub modules
  * ops.cpp_modules                 -> the reference's own C++ compiled into oracle/_ref
  * ops.fps.fps_utils               -> CPU FPS stand-in (oracle.fps); the CUDA op cannot run here, so
                                       the FPS index sequence itself is "parity unpinned"
  * torch.Tensor.cuda               -> identity
  * PointCloudSegment.ransac_plane_segmentation -> returns an injected model (Open3D is random)
  * np.bool = bool                  (alias removed from NumPy, used at compress_utils.py:157)
"""
import hashlib
import json
import os
import sys
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import oracle as orc  # noqa: E402
import rpcc_amd  # noqa: E402,F401
from rpcc_amd import synth  # noqa: E402


def sha(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.tobytes()).hexdigest()


def install_stubs():
    np.bool = bool

    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            for k, v in dict(d or {}, **kw).items():
                self[k] = v

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        __setattr__ = dict.__setitem__

    for name in ("IPython", "lz4", "open3d", "numba", "imageio"):
        sys.modules[name] = types.ModuleType(name)
    ed = types.ModuleType("easydict")
    ed.EasyDict = EasyDict
    sys.modules["easydict"] = ed

    ops = types.ModuleType("ops")
    ops.__path__ = []
    cppm = types.ModuleType("ops.cpp_modules")
    cppm.__path__ = [os.path.join(ROOT, "oracle", "_ref")]
    fpsp = types.ModuleType("ops.fps")
    fpsp.__path__ = []
    fu = types.ModuleType("ops.fps.fps_utils")

    def furthest_point_sample(xyz, npoint):
        idx = orc.fps(xyz[0].numpy(), npoint)
        return torch.from_numpy(idx.astype(np.int32))[None]

    fu.furthest_point_sample = furthest_point_sample
    sys.modules.update({"ops": ops, "ops.cpp_modules": cppm, "ops.fps": fpsp, "ops.fps.fps_utils": fu})
    ops.cpp_modules, ops.fps, fpsp.fps_utils = cppm, fpsp, fu
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)


def write_lidar_yaml(path, g):
    with open(path, "w") as f:
        f.write("HORIZONTAL_FOV: %r\nVERTICAL_ANGLE_MAX: %r\nVERTICAL_ANGLE_MIN: %r\n"
                "RANGE_IMAGE_HEIGHT: %d\nRANGE_IMAGE_WIDTH: %d\n" % (g["hfov_deg"], g["vmax_deg"], g["vmin_deg"], g["H"], g["W"]))


def estimate_ground(xyz):
    """A plain least-squares plane through the low points: only a plausible fp64 model to inject."""
    p = xyz[xyz[:, 2] < -1.3].astype(np.float64)
    c = p.mean(0)
    _, _, vt = np.linalg.svd(p - c, full_matrices=False)
    n = vt[-1]
    if n[2] > 0:
        n = -n
    return np.array([n[0], n[1], n[2], -n @ c])


def run_reference(xyz, lidar_yaml, ground_model, uniform, plane_rows=None, accuracy=0.02, cluster_num=None):
    """The body of tools/compress.py:44-131 driven through the genuine reference modules."""
    from dataset.transformer import PCTransformer
    from utils.segment_utils import PointCloudSegment
    from utils.compress_utils import QuantizationModule, compress_point_cloud, BasicCompressor
    import yaml

    cfg = yaml.safe_load(open(os.path.join(REF, "cfgs/compressor.yaml")))
    if cluster_num is not None:
        cfg["cluster_num"] = int(cluster_num)      # (the --cluster_num flag of tools/compress.py:20-37)
    acc = accuracy * 2
    T = PCTransformer(lidar_yaml, None)
    tm = T.transform_map
    ri = np.expand_dims(T.point_cloud_to_range_image(xyz), -1)
    pc = T.range_image_to_point_cloud(ri)
    seg_cfg = {"segment_method": cfg["segment_method"], "ground_vertical_threshold": cfg["ground_threshold"],
               "cluster_num": cfg["cluster_num"], "DBSCAN_eps": cfg["DBSCAN_eps"]}
    calls = {"n": 0}

    def fake_ransac(point_cloud, threshold=0.1, ransac_n=10, num_iterations=100):
        calls["n"] += 1
        if ransac_n == 10:
            return None, np.array(ground_model, dtype=np.float64)
        return None, np.array(plane_rows.pop(0), dtype=np.float64)

    PointCloudSegment.ransac_plane_segmentation = staticmethod(fake_ransac)
    ps = PointCloudSegment(tm)
    seg_idx, gm = ps.segment(pc, ri, seg_cfg, cpu=True)
    depth_dif = ps.calc_plane_residual_vertical(pc, gm)
    mask = depth_dif > seg_cfg["ground_vertical_threshold"]
    model_cfg = {"model_method": "plane" if plane_rows is not None else "point",
                 "angle_threshold": cfg["plane_angle_threshold"]}
    cluster_models = ps.cluster_modeling(pc, ri, seg_idx, model_cfg)
    model_param = np.concatenate((gm.reshape(1, 4), cluster_models), 0)
    pred = ps.intra_predict(seg_idx, model_param)
    residual = ri - pred
    if uniform:
        QM = QuantizationModule(acc)
    else:
        QM = QuantizationModule(acc, uniform=False, level_kp_num=tuple(cfg["level_key_point_num"]),
                                level_dacc=tuple(cfg["level_delta_acc"]),
                                ground_salience_level=cfg["ground_salience_level"],
                                feature_region=cfg["feature_region"], segments=cfg["segments"],
                                sharp_num=cfg["sharp_num"], less_sharp_num=cfg["less_sharp_num"],
                                flat_num=cfg["flat_num"])
        # the reference leaves key_point_map uninitialised (cpp_modules.cpp:38-43); the intended
        # zero-initialised semantics are obtained by masking with the positions it does write.
    q, sal, kp = QM.quantize_residual(residual, seg_idx, pc, ri)
    out = dict(tm=tm, ri=ri, pc=pc, mask=mask, seg_idx=seg_idx, model_param=model_param, pred=pred,
               residual=residual, q=q, salience=sal, key_point_map=kp)
    if uniform:
        bc = BasicCompressor(method_name="bzip2")
        od, cd = compress_point_cloud(bc, model_param, seg_idx, sal, q, pc, ri, full=False)
        out["payload"] = od
        parts = []
        import struct
        for k in ("contour_map", "idx_sequence", "plane_param", "residual_quantized"):
            parts += [struct.pack("i", len(cd[k])), cd[k]]
        out["rpcc"] = b"".join(parts)
    return out


SHA_GEOMS = ("Velodyne64E", "Velodyne64E_2048", "Velodyne32E", "VelodyneVLP16", "Velodyne64E_unofficial")   # every shipped lidar YAML + BASELINE's
SHA_FRAMES = tuple(range(100, 108))   # eight seeded sweeps per geometry


def sha_cases(tmp):
    """Breadth: eight seeded sweeps per geometry through the genuine reference, kept as SHA-256 digests only (no arrays): the inputs
    are regenerated from their seeds (synth.make_frame on the CPU; the digest of xyz guards that) -> manifest_sha.json."""
    lacc = np.array([0.04] * 4) + np.array([0, 0.02, 0.04, 0.06])
    from ops.cpp_modules import quantization_utils_cpp
    out = {"numpy": np.__version__, "glibc": os.confstr("CS_GNU_LIBC_VERSION"), "frames": list(SHA_FRAMES), "cases": {}}
    for geom in SHA_GEOMS:
        g = orc.GEOMS[geom]
        yml = os.path.join(tmp, geom + ".yaml")
        write_lidar_yaml(yml, g)
        rows = []
        for fid in SHA_FRAMES:
            xyz = synth.make_frame(fid, g["H"], g["W"], vmax_deg=g["vmax_deg"], vmin_deg=g["vmin_deg"]).numpy()
            gm = estimate_ground(xyz)
            ref = run_reference(xyz, yml, gm, uniform=True)
            # non-uniform framework with the zero-initialised key point map (see main()): the reference's own C++ on the oracle's map
            _, kp = orc.extract_features_with_segment(ref["ri"][..., 0], ref["seg_idx"])
            q_n, sal_n = quantization_utils_cpp.nonuniform_quantize(ref["seg_idx"], ref["residual"], kp, np.array([30, 10, 3, 0]), lacc, 2)
            rows.append(dict(frame=int(fid), n_points=int(xyz.shape[0]), nnz=int((ref["ri"] != 0).sum()), n_left=int(ref["mask"].sum()),
                             labels=int(ref["seg_idx"].max()) + 1, rpcc_bytes=len(ref["rpcc"]), ground_model=[float(v) for v in gm],
                             sha=dict(xyz=sha(xyz), ri=sha(ref["ri"]), mask=sha(ref["mask"]), seg_idx=sha(ref["seg_idx"].astype(np.uint8)),
                                      model_param=sha(ref["model_param"].astype(np.float32)), q=sha(ref["q"].astype(np.int16)),
                                      key_point_map=sha(kp.astype(np.uint8)), q_nonuniform=sha(q_n.astype(np.int16)),
                                      salience=sha(sal_n.astype(np.uint8)), rpcc=hashlib.sha256(ref["rpcc"]).hexdigest())))
            print(geom, fid, rows[-1]["nnz"], rows[-1]["n_left"], rows[-1]["rpcc_bytes"], flush=True)
        out["cases"][geom] = rows
    # cluster_num above 254 (labels beyond a byte; the reference writes them as uint16, utils/compress_utils.py:160): two VLP-16 sweeps at 300 clusters
    geom, M = "VelodyneVLP16", 300
    g = orc.GEOMS[geom]
    yml = os.path.join(tmp, geom + ".yaml")
    rows = []
    for fid in (100, 101):
        xyz = synth.make_frame(fid, g["H"], g["W"], vmax_deg=g["vmax_deg"], vmin_deg=g["vmin_deg"]).numpy()
        gm = estimate_ground(xyz)
        ref = run_reference(xyz, yml, gm, uniform=True, cluster_num=M)
        assert int(ref["seg_idx"].max()) == M + 1
        rows.append(dict(frame=int(fid), cluster_num=M, nnz=int((ref["ri"] != 0).sum()), labels=int(ref["seg_idx"].max()) + 1, rpcc_bytes=len(ref["rpcc"]),
                         ground_model=[float(v) for v in gm],
                         sha=dict(xyz=sha(xyz), seg_idx=sha(ref["seg_idx"].astype(np.uint16)), model_param=sha(ref["model_param"].astype(np.float32)),
                                  q=sha(ref["q"].astype(np.int16)), rpcc=hashlib.sha256(ref["rpcc"]).hexdigest())))
        print("wide", geom, fid, rows[-1]["nnz"], rows[-1]["rpcc_bytes"], flush=True)
    out["wide"] = {"geom": geom, "rows": rows}
    json.dump(out, open(os.path.join(HERE, "manifest_sha.json"), "w"), indent=1, sort_keys=True)


def main(argv):
    """python gen_golden.py [full] [kitti_test] [sha]   (default: all; `kitti_test` alone adds that case to the committed manifest)"""
    what = set(argv) or {"full", "sha"}
    install_stubs()
    tmp = "/tmp/rpcc_golden_tmp"
    os.makedirs(tmp, exist_ok=True)
    if "sha" in what:
        sha_cases(tmp)
    if not (what & {"full", "kitti_test"}):
        return
    manifest = {"numpy": np.__version__, "glibc": os.confstr("CS_GNU_LIBC_VERSION"), "cases": {}}
    if "full" not in what:   # add to the committed manifest without regenerating the other fixtures
        manifest = json.load(open(os.path.join(HERE, "manifest.json")))

    cases = []
    if "full" in what:
        ex = np.fromfile(os.path.join(REF, "assets/example_data/example.bin"), dtype=np.float32).reshape(-1, 4)[:, :3]
        cases.append(("example_64E", "Velodyne64E", np.ascontiguousarray(ex),
                      np.array([0.00721658, -0.0544943, -0.998488, -1.75882636])))
    for name, geom, fid in (("synth_64x2048", "Velodyne64E_2048", 0), ("synth_32E", "Velodyne32E", 1),
                            ("synth_vlp16", "VelodyneVLP16", 2), ("synth_kitti_test", "Velodyne64E_unofficial", 3)):
        if "full" not in what and name != "synth_kitti_test":
            continue
        g = orc.GEOMS[geom]
        xyz = synth.make_frame(fid, g["H"], g["W"], vmax_deg=g["vmax_deg"], vmin_deg=g["vmin_deg"]).numpy()
        cases.append((name, geom, xyz, estimate_ground(xyz)))

    for name, geom, xyz, gm in cases:
        g = orc.GEOMS[geom]
        yml = os.path.join(tmp, geom + ".yaml")
        write_lidar_yaml(yml, g)
        ref_u = run_reference(xyz, yml, gm, uniform=True)
        ref_n = run_reference(xyz, yml, gm, uniform=False)
        # zero-init semantics for the key point map: keep only positions where a feature was written
        # this run.  Re-derive with the oracle restatement and require agreement on written cells.
        feat_o, kp_o = orc.extract_features_with_segment(ref_n["ri"][..., 0], ref_n["seg_idx"])
        lacc = np.array([0.04] * 4) + np.array([0, 0.02, 0.04, 0.06])
        q_n, sal_n = orc.nonuniform_quantize(ref_n["seg_idx"], ref_n["residual"], kp_o, np.array([30, 10, 3, 0]), lacc, 2)
        # the genuine C++ with the zero-initialised key point map:
        from ops.cpp_modules import quantization_utils_cpp
        q_ref_n, sal_ref_n = quantization_utils_cpp.nonuniform_quantize(
            ref_n["seg_idx"], ref_n["residual"], kp_o, np.array([30, 10, 3, 0]), lacc, 2)
        assert np.array_equal(q_n, q_ref_n) and np.array_equal(sal_n, sal_ref_n)

        np.savez_compressed(os.path.join(HERE, name + ".npz"),
                            xyz=xyz, ground_model=gm,
                            seg_idx=ref_u["seg_idx"].astype(np.uint8),
                            model_param=ref_u["model_param"],
                            q_uniform=ref_u["q"].astype(np.int16),
                            q_nonuniform=q_ref_n.astype(np.int16), salience=sal_ref_n.astype(np.uint8),
                            key_point_map=kp_o.astype(np.uint8),
                            rpcc=np.frombuffer(ref_u["rpcc"], dtype=np.uint8))
        manifest["cases"][name] = dict(
            geom=geom, n_points=int(xyz.shape[0]),
            nnz=int((ref_u["ri"] != 0).sum()), n_left=int(ref_u["mask"].sum()),
            sha=dict(tm=sha(ref_u["tm"]), ri=sha(ref_u["ri"]), pc=sha(ref_u["pc"]), mask=sha(ref_u["mask"]),
                     seg_idx=sha(ref_u["seg_idx"].astype(np.int32)), model_param=sha(ref_u["model_param"]),
                     pred=sha(ref_u["pred"]), residual=sha(ref_u["residual"]), q=sha(ref_u["q"].astype(np.int32)),
                     contour_map=sha(ref_u["payload"]["contour_map"]), idx_sequence=sha(ref_u["payload"]["idx_sequence"]),
                     rpcc=hashlib.sha256(ref_u["rpcc"]).hexdigest()),
            q_range=[int(ref_u["q"].min()), int(ref_u["q"].max())], rpcc_bytes=len(ref_u["rpcc"]))
        print(name, manifest["cases"][name]["nnz"], manifest["cases"][name]["n_left"], len(ref_u["rpcc"]))

    # the only known-answer vector the reference itself holds: utils/contour_utils.py:181-195
    manifest["contour_kat"] = dict(idx_map=[[1, 1, 1, 1, 2], [3, 2, 2, 1, 2], [3, 2, 1, 1, 2], [3, 3, 2, 2, 2]],
                                   contour=[[1, 0, 0, 0, 1], [1, 1, 0, 1, 1], [1, 1, 1, 0, 1], [1, 0, 1, 0, 0]],
                                   idx_sequence=[1, 2, 3, 2, 1, 2, 3, 2, 1, 2, 3, 2])
    json.dump(manifest, open(os.path.join(HERE, "manifest.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1:])

"""CPU ORACLE (test infrastructure) for the R-PCC hot path.

ctypes front-end of oracle/liborpcc.so (rpcc_oracle.c) plus NumPy restatements of the rows of the
reference that are Python (a1, a3, a5, a7 and the glue of a8/a11) and the
host-side payload packing (f1, f2).  Only tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke() may import this module; the product package never does.

All `path:line` citations are relative to /root/reference.
"""
import bz2
import ctypes as C
import math
import os
import struct
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile liborpcc.so with gcc (and oracle/_ref when /root/reference is present)."""
    so = os.path.join(_HERE, "liborpcc.so")
    src = os.path.join(_HERE, "rpcc_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liborpcc.so"])
    if os.path.exists("/root/reference/ops/cpp_modules/src/cpp_modules.cpp"):
        if force or not os.path.exists(os.path.join(_HERE, "_ref", ".built")):
            subprocess.check_call(["make", "-C", _HERE, "-s", "ref"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_atan2f.restype = C.c_float
        _LIB.orc_atan2f.argtypes = [C.c_float, C.c_float]
        _LIB.orc_plane_divisor4.restype = C.c_double
        _LIB.orc_np_mean_f32.restype = C.c_float
        _LIB.orc_np_mean_f32.argtypes = [C.c_void_p, C.c_long]
        for n in ("orc_uniform_quantize", "orc_nonuniform_quantize", "orc_extract_contour", "orc_ransac_plane",
                  "orc_ground_candidates"):
            getattr(_LIB, n).restype = C.c_long
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


# ------------------------------------------------------------------------------------------------
# lidar geometry + a1 transform map (dataset/transformer.py:26-54)
# ------------------------------------------------------------------------------------------------
class LidarGeom:
    """The scalars PCTransformer.__init__ derives from a lidar YAML (dataset/transformer.py:26-37)."""

    def __init__(self, H, W, hfov_deg=360.0, vmax_deg=2.0, vmin_deg=-24.9):
        self.H, self.W = int(H), int(W)
        self.horizontal_FOV = hfov_deg * (np.pi / 180)
        self.vertical_max = vmax_deg * (np.pi / 180)
        self.vertical_min = vmin_deg * (np.pi / 180)
        self.vertical_FOV = self.vertical_max - self.vertical_min


GEOMS = {
    "Velodyne64E": dict(H=64, W=2000, hfov_deg=360, vmax_deg=2.0, vmin_deg=-24.9),
    "Velodyne64E_2048": dict(H=64, W=2048, hfov_deg=360, vmax_deg=2.0, vmin_deg=-24.9),
    "Velodyne32E": dict(H=32, W=2250, hfov_deg=360, vmax_deg=10.67, vmin_deg=-30.67),
    "VelodyneVLP16": dict(H=16, W=1800, hfov_deg=360, vmax_deg=15, vmin_deg=-15),
    "Velodyne64E_unofficial": dict(H=80, W=2000, hfov_deg=360, vmax_deg=2.5, vmin_deg=-23.6),   # dataset/__init__.py:21 'KITTI_test'
}


def transform_map(g):
    """a1: create_transform_map, dataset/transformer.py:41-54 -- python-float (fp64) cos/sin
    products, then astype(float32).  math.cos/math.sin are kept (NumPy's SIMD cos may differ in the
    last ulp); the double loop is replaced by an outer product of the same fp64 factors."""
    ca = np.array([math.cos(g.vertical_FOV * (h / (g.H - 1)) + g.vertical_min) for h in range(g.H)])
    sa = np.array([math.sin(g.vertical_FOV * (h / (g.H - 1)) + g.vertical_min) for h in range(g.H)])
    cz = np.array([math.cos(g.horizontal_FOV * (w / g.W)) for w in range(g.W)])
    sz = np.array([math.sin(g.horizontal_FOV * (w / g.W)) for w in range(g.W)])
    tm = np.zeros((g.H, g.W, 3))
    tm[..., 0] = ca[:, None] * cz[None, :]
    tm[..., 1] = ca[:, None] * sz[None, :]
    tm[..., 2] = sa[:, None]
    return tm.astype(np.float32)


# ------------------------------------------------------------------------------------------------
# C-backed stage functions
# ------------------------------------------------------------------------------------------------
def project(xyz, g):
    """a2: point_cloud_to_range_image_even (cpp_modules.cpp:427-467); python doubles narrowed to C float."""
    xyz = _f32(xyz)
    ri = np.empty((g.H, g.W), np.float32)
    lib().orc_project(_p(xyz), C.c_long(xyz.shape[0]), g.H, g.W, C.c_float(g.horizontal_FOV),
                      C.c_float(g.vertical_max), C.c_float(g.vertical_min), _p(ri))
    return ri


def backproject(ri, tm):
    """a3: range_image_to_point_cloud (dataset/transformer.py:94-101)."""
    return ri.reshape(ri.shape[0], ri.shape[1], 1).astype(np.float32) * tm


def vertical_residual(pc, plane):
    """a5: calc_plane_residual_vertical cpu branch (utils/segment_utils.py:44-47), fp64."""
    pc = _f32(pc)
    plane = np.ascontiguousarray(plane, np.float64)
    out = np.empty(pc.shape[:-1], np.float64)
    lib().orc_vertical_residual(_p(pc), _p(plane), C.c_long(out.size), _p(out))
    return out


def np_vertical_residual(pc, plane):
    """Same row written with the reference's own NumPy expression (incl. the [:, :3] slice on a
    (1,1,4) array that keeps four components in the divisor)."""
    pp = np.expand_dims(np.expand_dims(np.asarray(plane, np.float64), 0), 0)
    return np.abs(np.sum(pc * pp[..., :3], -1) + pp[..., 3]) / np.linalg.norm(pp[:, :3], 2, -1)


def fps(xyz, m):
    """a6: furthest_point_sample (ops/fps/src/sampling_gpu.cu:44-69,136-138), sequential restatement."""
    xyz = _f32(xyz)
    idx = np.zeros(m, np.int32)
    lib().orc_fps(_p(xyz), int(xyz.shape[0]), int(m), _p(idx))
    return idx


def fps_modes(xyz, m, fma=0, cuda_tie=False):
    """a6 with the CUDA binary's two unpinned degrees of freedom as switches (orc_fps_modes): fma = 0 un-fused,
    1 fma(dz,dz,fma(dx,dx,dy*dy)), 2 fma(dz,dz,fma(dy,dy,dx*dx)) -- nvcc's --fmad=true contractions of sampling_gpu.cu:64;
    cuda_tie: the survivor of the kernel's own reduction among exactly equal values (sampling_gpu.cu:16-21,55-69,74-134)
    instead of the lowest index.  fps_modes(xyz, m) == fps(xyz, m)."""
    xyz = _f32(xyz)
    idx = np.zeros(m, np.int32)
    lib().orc_fps_modes(_p(xyz), int(xyz.shape[0]), int(m), int(fma), int(bool(cuda_tie)), _p(idx))
    return idx


def fps_cuda_emulated(xyz, m, fma=0):
    """The CUDA kernel thread by thread (strided per-thread scan + shared-memory tree): small inputs only (test of the
    closed-form tie rule of fps_modes(cuda_tie=True))."""
    xyz = _f32(xyz)
    idx = np.zeros(m, np.int32)
    lib().orc_fps_cuda_emulated(_p(xyz), int(xyz.shape[0]), int(m), int(fma), _p(idx))
    return idx


def assign(ri, pc, tm, plane, centers):
    """a7: ground/cluster arg-nearest + relabel (utils/segment_utils.py:127-131,168-169) -> int32 [H,W]."""
    ri2 = _f32(ri).reshape(pc.shape[0], pc.shape[1])
    pc, tm, centers = _f32(pc), _f32(tm), _f32(centers)
    plane = np.ascontiguousarray(plane, np.float64)
    seg = np.empty(ri2.shape, np.int32)
    lib().orc_assign(_p(ri2), _p(pc), _p(tm), _p(plane), _p(centers), int(centers.shape[0]),
                     C.c_long(seg.size), _p(seg))
    return seg


def np_assign(ri, pc, tm, plane, centers):
    """a7 written with the reference's own NumPy expressions (slow; small cases only)."""
    h, w, c = pc.shape
    pp = np.expand_dims(np.expand_dims(np.asarray(plane, np.float64), 0), 0)
    r_plane = -pp[..., 3] / np.sum(pp[..., :3] * tm, -1)
    ground_residual = ri.reshape(h, w) - r_plane
    diff = np.reshape(pc, (h, w, 1, c)) - np.reshape(centers, (1, 1, centers.shape[0], c))
    radius = np.linalg.norm(diff, 2, -1)
    distance = np.concatenate((np.expand_dims(ground_residual, -1), radius), -1)
    seg = np.argmax(-np.abs(distance), -1)
    seg[np.where(seg > 0)] += 1
    seg[np.where(ri.reshape(h, w) == 0)] = 1
    return seg


def point_modeling(ri, seg):
    """a8: point_modeling (cpp_modules.cpp:471-518) -> f32[max+1]."""
    ri, seg = _f32(ri), _i32(seg)
    out = np.empty(int(seg.max()) + 1, np.float32)
    lib().orc_point_modeling(_p(ri), _p(seg), C.c_long(seg.size), _p(out))
    return out


def intra_predict(seg, model_param, tm):
    """a10: intra_predict (cpp_modules.cpp:248-285); model_param is cast to fp32 like pybind11 does."""
    seg, mp, tm = _i32(seg), _f32(model_param), _f32(tm)
    pred = np.empty(seg.shape + (1,), np.float32)
    lib().orc_intra_predict(_p(seg), _p(mp), _p(tm), C.c_long(seg.size), _p(pred))
    return pred


def uniform_quantize(seg, residual, acc):
    """a11: uniform_quantize (cpp_modules.cpp:288-334) -> int32[nnz]."""
    seg, residual = _i32(seg), _f32(residual)
    out = np.empty(seg.size, np.int32)
    n = lib().orc_uniform_quantize(_p(seg), _p(residual), C.c_long(seg.size), C.c_float(acc), _p(out))
    return out[:n].copy()


def extract_features_with_segment(ri, seg, feature_region=3, segments=8, sharp_num=4, less_sharp_num=8, flat_num=6):
    """a12: extract_features_with_segment (cpp_modules.cpp:28-121), zero-initialised outputs."""
    ri, seg = _f32(ri), _i32(seg)
    h, w = seg.shape
    feat = np.empty((h, w), np.float32)
    kp = np.empty((h, w), np.int32)
    lib().orc_extract_features_with_segment(_p(ri), _p(seg), h, w, feature_region, segments, sharp_num,
                                            less_sharp_num, flat_num, _p(feat), _p(kp))
    return feat, kp


def nonuniform_quantize(seg, residual, kp, level_kp_num, level_acc, ground_level):
    """a13: nonuniform_quantize (cpp_modules.cpp:337-424) -> (int32[nnz], int32[max+1])."""
    seg, residual, kp = _i32(seg), _f32(residual), _i32(kp)
    lk, la = _i32(level_kp_num), _f32(level_acc)
    out = np.empty(seg.size, np.int32)
    sal = np.empty(int(seg.max()) + 1, np.int32)
    n = lib().orc_nonuniform_quantize(_p(seg), _p(residual), _p(kp), _p(lk), _p(la), int(la.shape[0]),
                                      int(ground_level), C.c_long(seg.size), _p(out), _p(sal))
    return out[:n].copy(), sal


def extract_contour(idx_map):
    """f1: extract_contour (cpp_modules.cpp:521-558)."""
    im = _i32(idx_map)
    h, w = im.shape
    cm = np.empty((h, w), np.int32)
    seq = np.empty(h * w, np.int32)
    n = lib().orc_extract_contour(_p(im), h, w, _p(cm), _p(seq))
    return cm, seq[:n].copy()


def recover_map(contour_map, idx_sequence):
    """f3: recover_map (cpp_modules.cpp:561-593)."""
    cm, seq = _i32(contour_map), _i32(idx_sequence)
    h, w = cm.shape
    im = np.zeros((h, w), np.int32)
    lib().orc_recover_map(_p(cm), _p(seq), C.c_long(seq.shape[0]), h, w, _p(im))
    return im


def np_mean_f32(a):
    a = _f32(a).reshape(-1)
    return np.float32(lib().orc_np_mean_f32(_p(a), C.c_long(a.size)))


def ransac_plane(pts, ransac_n, iters, thr=0.1, seed=0):
    """Build-defined seeded RANSAC (NOT Open3D): sequential form of the specification the HIP kernel
    implements (DESIGN.md "RANSAC").  -> (plane fp64[4], inlier count of the winning hypothesis)."""
    pts = _f32(pts)
    plane = np.zeros(4, np.float64)
    n = lib().orc_ransac_plane(_p(pts), C.c_long(pts.shape[0]), int(ransac_n), int(iters), C.c_double(thr),
                               C.c_uint32(seed), _p(plane))
    return plane, int(n)


def ground_candidates(ri, tm, zthr=-1.5, max_pts=5000, min_pts=800):
    """Ground RANSAC input (utils/segment_utils.py:101-106) with the deterministic subsample."""
    ri, tm = _f32(ri).reshape(-1), _f32(tm)
    out = np.empty((ri.size, 3), np.float32)
    m = lib().orc_ground_candidates(_p(ri), _p(tm), C.c_long(ri.size), C.c_float(zthr), C.c_long(max_pts),
                                    C.c_long(min_pts), _p(out))
    return out[:m].copy()


def ground_model(ri, tm, seed=0):
    """a4 with the build's seeded RANSAC: candidates -> plane (ransac_n=10, 100 iterations, 0.1 m)."""
    return ransac_plane(ground_candidates(ri, tm), 10, 100, 0.1, seed)[0]


def mix32(a, b, c):
    lib().orc_mix32.restype = C.c_uint32
    return int(lib().orc_mix32(C.c_uint32(a & 0xFFFFFFFF), C.c_uint32(b & 0xFFFFFFFF), C.c_uint32(c & 0xFFFFFFFF)))


def plane_angle_ok(plane_model, scan_vector, angle_deg):
    """plane_angle_validation (utils/segment_utils.py:84-93), the reference's own numpy expression."""
    with np.errstate(invalid="ignore"):
        alpha = np.arccos(np.abs(np.sum(np.expand_dims(plane_model[:3], 0) * scan_vector, -1)) /
                          np.linalg.norm(plane_model[:3]) * np.linalg.norm(scan_vector, ord=2, axis=-1))
    return not (alpha.max() > np.pi * (angle_deg / 180))


def cluster_modeling_plane(pc, ri, seg, tm, angle_deg=75, seed=0, frame=0, inject=None):
    """a9: cluster_modeling('plane') (utils/segment_utils.py:188-216) with the build's seeded RANSAC in
    place of Open3D: label k of frame b uses seed mix32(seed, b, k).  -> float64 [max, 4] (rows for labels 1..max).
    inject: list of plane rows handed out, in call order, INSTEAD of the RANSAC result (what gen_golden_pins.py does to
    the genuine reference through ransac_plane_segmentation): pins the glue around the fit."""
    rows = []
    inject = None if inject is None else [np.asarray(r, np.float64) for r in inject]
    ri3 = ri.reshape(seg.shape[0], seg.shape[1], 1)
    for i in range(int(seg.max()) + 1):
        if i == 0:
            continue
        if i == 1:
            rows.append([0, 0, 0, 0.0])
            continue
        idx = np.where(seg == i)
        cur = ri3[idx]
        if idx[0].shape[0] < 30:
            rows.append([0, 0, 0, np_mean_f32(cur)])
            continue
        if inject is not None:
            plane = inject.pop(0)
        else:
            plane, _ = ransac_plane(pc[idx], 4, 10, 0.1, mix32(seed, frame, i))
        if plane_angle_ok(plane, tm[idx], angle_deg):
            rows.append(list(plane))
        else:
            rows.append([0, 0, 0, np_mean_f32(cur)])
    return np.asarray(rows)


# ------------------------------------------------------------------------------------------------
# Per-frame pipeline glue (tools/compress.py:93-131) with injected ground / plane models
# ------------------------------------------------------------------------------------------------
DEFAULT_CFG = dict(accuracy=0.02, ground_threshold=0.1, cluster_num=100, level_key_point_num=(30, 10, 3, 0),
                   level_delta_acc=(0, 0.02, 0.04, 0.06), ground_salience_level=2, feature_region=3, segments=8,
                   sharp_num=4, less_sharp_num=8, flat_num=6)


def segment(ri, tm, ground_model, cfg=DEFAULT_CFG):
    """segment() cpu=True branch after the ground RANSAC (utils/segment_utils.py:119-131,168-169).
    Returns dict(pc, depth_dif, mask, fps_idx (into the compacted list), fps_pix (pixel index),
    centers, seg_idx int64)."""
    H, W = ri.shape[:2]
    pc = backproject(ri, tm)
    dd = vertical_residual(pc, ground_model)
    mask = dd > cfg["ground_threshold"]
    pc_left = pc[np.where(mask)]
    if cfg.get("fps_fma", 0) or cfg.get("fps_cuda_tie", False):
        idx = fps_modes(pc_left, cfg["cluster_num"], cfg.get("fps_fma", 0), cfg.get("fps_cuda_tie", False))
    else:
        idx = fps(pc_left, cfg["cluster_num"])
    centers = pc_left[idx]
    pix = np.flatnonzero(mask.reshape(-1))[idx].astype(np.int32)
    seg = assign(ri, pc, tm, ground_model, centers).astype(np.int64)
    return dict(pc=pc, depth_dif=dd, mask=mask, fps_idx=idx, fps_pix=pix, centers=centers, seg_idx=seg)


def point_model_param(ri, seg, ground_model):
    """cluster_modeling('point') + model_param concat (segment_utils.py:182-185, compress.py:102)."""
    cm = point_modeling(ri, seg)
    cm = np.concatenate((np.zeros((cm.shape[0], 3)), np.expand_dims(cm, -1)), -1)[1:]
    return np.concatenate((np.asarray(ground_model, np.float64).reshape(1, 4), cm), 0)


def compress_frame(xyz, g, tm, ground_model, cfg=DEFAULT_CFG, uniform=True, model_param=None, plane=None):
    """Whole hot path for one frame (tools/compress.py:93-125), ground model injected.  When
    `model_param` is given it replaces cluster_modeling (plane-mode injection); plane = dict(angle_deg, seed, frame)
    selects cluster_modeling('plane') with the build's seeded RANSAC (cluster_modeling_plane)."""
    ri = project(xyz, g)
    s = segment(ri, tm, ground_model, cfg)
    seg = s["seg_idx"]
    if model_param is None and plane is not None:
        model_param = np.concatenate((np.asarray(ground_model, np.float64).reshape(1, 4),
                                      cluster_modeling_plane(s["pc"], ri, seg, tm, plane["angle_deg"], plane["seed"], plane["frame"])), 0)
    if model_param is None:
        model_param = point_model_param(ri, seg, ground_model)
    pred = intra_predict(seg, model_param, tm)
    residual = ri.reshape(g.H, g.W, 1) - pred
    acc = cfg["accuracy"] * 2
    out = dict(range_image=ri, model_param=model_param, pred=pred, residual=residual, **s)
    if uniform:
        out["q"] = uniform_quantize(seg, residual, acc)
        out["salience"] = None
        out["key_point_map"] = None
    else:
        feat, kp = extract_features_with_segment(ri, seg, cfg["feature_region"], cfg["segments"], cfg["sharp_num"],
                                                 cfg["less_sharp_num"], cfg["flat_num"])
        lacc = np.array([acc] * len(cfg["level_key_point_num"])) + np.array(cfg["level_delta_acc"])
        q, sal = nonuniform_quantize(seg, residual, kp, np.array(cfg["level_key_point_num"]), lacc,
                                     cfg["ground_salience_level"])
        out.update(q=q, salience=sal, key_point_map=kp, feat=feat)
    return out


# ------------------------------------------------------------------------------------------------
# f1/f2: payload packing + container (utils/compress_utils.py:138-179), bzip2 back-end
# ------------------------------------------------------------------------------------------------
def pack_payload(model_param, seg, salience, q):
    cm, seq = extract_contour(seg)
    od = {"residual_quantized": q.astype(np.int16)}
    if salience is not None:
        od["salience_level"] = salience.astype(np.uint8)
    od["contour_map"] = np.packbits(cm.astype(bool), axis=None).astype(np.uint8)
    od["idx_sequence"] = seq.astype(np.uint16)
    od["plane_param"] = np.asarray(model_param).astype(np.float32)
    return od


def bitstream_bytes(od, uniform=True):
    comp = {k: bz2.compress(v) for k, v in od.items()}
    parts = []
    keys = ([] if uniform else ["salience_level"]) + ["contour_map", "idx_sequence", "plane_param", "residual_quantized"]
    for k in keys:
        parts.append(struct.pack("i", len(comp[k])))
        parts.append(comp[k])
    return b"".join(parts)


# ------------------------------------------------------------------------------------------------
# f3: decoder (utils/compress_utils.py:199-214 decompress_point_cloud, :114-132 dequantize_residual,
#     tools/decompress.py:88-112)
# ------------------------------------------------------------------------------------------------
def unpack_bitstream(blob, uniform=True):
    """read_compressed_bitstream (utils/compress_utils.py:181-196) on bytes + bz2 -> dict of raw byte strings."""
    keys = ([] if uniform else ["salience_level"]) + ["contour_map", "idx_sequence", "plane_param", "residual_quantized"]
    out, off = {}, 0
    for k in keys:
        (n,) = struct.unpack("i", blob[off:off + 4])
        out[k] = bz2.decompress(blob[off + 4:off + 4 + n])
        off += 4 + n
    assert off == len(blob)
    return out


def decode_frame(blob, g, tm, accuracy=0.02, uniform=True, level_delta_acc=(0, 0.02, 0.04, 0.06)):
    """The reference decoder on one .rpcc byte string -> dict(seg_idx, residual f32 [H,W,1], pred, ri_rec, pc_rec)."""
    d = unpack_bitstream(blob, uniform)
    plane_param = np.frombuffer(d["plane_param"], dtype=np.float32).reshape(-1, 4)   # all rows (the reference's shape
    #                                  (cluster_num+1, 4) is one row short but aliases the same buffer: SURVEY 8c)
    contour = np.unpackbits(np.frombuffer(d["contour_map"], dtype=np.uint8))[: g.H * g.W].reshape(g.H, g.W)
    seq = np.frombuffer(d["idx_sequence"], dtype=np.uint16)
    seg = recover_map(contour.astype(np.int32), seq.astype(np.int32))
    rq = np.frombuffer(d["residual_quantized"], dtype=np.int16)
    step = accuracy * 2                                   # tools/decompress.py:58: python float
    acc = step if uniform else np.array([step] * len(level_delta_acc)) + np.array(level_delta_acc)
    sal = None if uniform else np.frombuffer(d["salience_level"], dtype=np.uint8)
    residual = np.zeros(seg.shape, dtype=np.float32)      # compress_utils.py:115
    start = 0
    for m in range(int(seg.max()) + 1):
        idx = np.where(seg == m)
        if m == 1:
            continue
        cur = acc if uniform else acc[sal[m]]
        residual[idx] = rq[start:start + idx[0].shape[0]] * cur     # int16 * python float / np.float64 -> fp64, stored fp32
        start += idx[0].shape[0]
    assert start == rq.shape[0]
    residual = np.expand_dims(residual, -1)
    pred = intra_predict(seg, plane_param.astype(np.float64), tm)
    ri_rec = pred + residual
    pc_rec = ri_rec * tm                                   # dataset/transformer.py:94-101
    return dict(seg_idx=seg, residual=residual, pred=pred, ri_rec=ri_rec, pc_rec=pc_rec.astype(np.float32))

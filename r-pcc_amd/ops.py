"""Stage-level Python wrappers of the C ABI (include/rpcc_hip.h): torch tensors in, torch tensors out.

PyTorch is only the device-memory container and the stream provider here; all compute is in
librpcc_hip.so.  Every function enqueues on torch's current HIP stream and does not synchronise.
"""
import ctypes as C
import functools
import math
import os

import numpy as np
import torch

from . import _lib
from ._lib import Geom, BatchIO, check, ptr, stream

DEFAULT_CLUSTERS = 100


def make_geom(H, W, horizontal_FOV, vertical_max, vertical_min):
    """Geometry scalars as pybind11 hands them to the reference C++ (python doubles narrowed to float,
    cpp_modules.cpp:427-428)."""
    return Geom(int(H), int(W), float(horizontal_FOV), float(vertical_max), float(vertical_min))


def transform_map(H, W, horizontal_FOV, vertical_max, vertical_min):
    """a1: PCTransformer.create_transform_map (dataset/transformer.py:41-54).  Frame-invariant host
    work (once per process): python-float cos/sin products rounded to fp32; the reference's H*W Python
    loop is evaluated as an outer product of the same fp64 factors."""
    vfov = vertical_max - vertical_min
    alt = [vfov * (h / (H - 1)) + vertical_min for h in range(H)]
    azi = [horizontal_FOV * (w / W) for w in range(W)]
    ca = np.array([math.cos(a) for a in alt]); sa = np.array([math.sin(a) for a in alt])
    cz = np.array([math.cos(a) for a in azi]); sz = np.array([math.sin(a) for a in azi])
    tm = np.zeros((H, W, 3))
    tm[..., 0] = ca[:, None] * cz[None, :]
    tm[..., 1] = ca[:, None] * sz[None, :]
    tm[..., 2] = sa[:, None]
    return tm.astype(np.float32)


def _dev(t):
    return t.device


def _point_stride(xyz):
    """Bytes per point of a point tensor: [N,3] packed xyz -> 12; [N,4] rows (x, y, z, intensity) as a KITTI .bin stores them
    (dataset/dataset.py:48-50) -> 16, read by the kernel as they are."""
    assert xyz.dim() == 2 and xyz.shape[1] in (3, 4) and xyz.dtype == torch.float32, "points: f32 [N,3] or [N,4]"
    return 4 * int(xyz.shape[1])


PROJECT_ORDER_PROBE, PROJECT_FORCE_ORDERED = 16, 32    # include/rpcc_hip.h: RPCC_PROJECT_*


def _project_flags(flags):
    """The caller's RPCC_PROJECT_* bits | the environment's (RPCC_PROJECT_FLAGS=16: probe the point order of every frame and project the frames in
    scanner order without records, 32: every frame through that kernel -- like RPCC_FPS_FMA a process-wide switch that every front-end honours;
    results do not depend on it)."""
    return (int(flags) | int(os.environ.get("RPCC_PROJECT_FLAGS", "0") or 0)) & (PROJECT_ORDER_PROBE | PROJECT_FORCE_ORDERED)


def project(xyz, offsets, geom, ri=None, scratch=None, atomic_path=False, order_flags=0, accepted=None):
    """a2 batched.  xyz f32 [total,3] -- or the stored rows f32 [total,4] (x, y, z, intensity), read with a 16-byte stride --,
    offsets i64 [B+1] (device) -> ri f32 [B,H,W].
    atomic_path=True gives the library only the small scratch, which selects the device-atomic kernels
    (same result as the default LDS-band kernels).  order_flags: 0, PROJECT_ORDER_PROBE (a frame whose points come in scanner order takes
    the record-free window kernel, chosen by a probe) or PROJECT_FORCE_ORDERED (rpcc_project_ordered); accepted: i32 [B] tensor that
    receives which frames the window kernel took."""
    B = offsets.numel() - 1
    P = geom.H * geom.W
    xyz = xyz.contiguous()
    if ri is None:
        ri = torch.empty((B, geom.H, geom.W), dtype=torch.float32, device=_dev(offsets))
    if scratch is None:
        n = B * (P + 8) * 4 if atomic_path else _lib.lib().rpcc_project_scratch_bytes(xyz.shape[0], B, P)
        scratch = torch.empty(n, dtype=torch.uint8, device=_dev(offsets))
    check(_lib.lib().rpcc_project_ordered(ptr(xyz) if xyz.numel() else None, _point_stride(xyz), ptr(offsets), xyz.shape[0], B, geom,
                                          ptr(ri), ptr(scratch), scratch.numel() * scratch.element_size(), _project_flags(order_flags),
                                          ptr(accepted) if accepted is not None else None, stream()))
    return ri


def project_fastpath_check(xyz, geom):
    """Test hook of a2: (points the screened fast path is certain about, of those the ones whose pixel differs
    from the exact sequence -- must be 0 --, points sent to the exact sequence, max column discrepancy, max row
    discrepancy of the pre-rounding coordinates [pixels])."""
    counts = torch.zeros((5,), dtype=torch.int64, device=_dev(xyz))
    check(_lib.lib().rpcc_project_fastpath_check(ptr(xyz), int(xyz.shape[0]), geom, ptr(counts), stream()))
    c = [int(v) for v in counts.cpu()]
    return c[0], c[1], c[2], c[3] * 1e-9, c[4] * 1e-9


def _frame_ids(frame_ids, B, device):
    """Stable per-frame identities (datalist indices) as a device i64 [B] tensor, or None."""
    if frame_ids is None:
        return None
    if not torch.is_tensor(frame_ids):
        frame_ids = torch.as_tensor(np.asarray(frame_ids, dtype=np.int64))
    frame_ids = frame_ids.to(device=device, dtype=torch.int64).contiguous()
    assert frame_ids.numel() == B, "one identity per frame"
    return frame_ids


def ground_ransac(ri, tm, seed=0, frame_ids=None):
    """a4: seeded ground-plane RANSAC -> (ground f64 [B,4], inlier counts i32 [B]).  Frame b draws with
    seed + frame_ids[b] (its datalist index; default: its position b in the batch)."""
    B = ri.shape[0]
    P = ri[0].numel()
    ground = torch.empty((B, 4), dtype=torch.float64, device=_dev(ri))
    inl = torch.empty((B,), dtype=torch.int32, device=_dev(ri))
    fid = _frame_ids(frame_ids, B, _dev(ri))
    check(_lib.lib().rpcc_ground_ransac(ptr(ri), ptr(tm), B, P, int(seed) & 0xFFFFFFFF, ptr(fid), ptr(ground), ptr(inl),
                                        stream()))
    return ground, inl


def ground_mask(ri, tm, ground, threshold, fps_table=False):
    """a3+a5.  -> (temp f32 [B,P], info i32 [B,8] = n_left, first candidate pixel, nnz, table flag, first empty
    candidate pixel, 3 spare).
    fps_table=True: the kernel also runs the first FPS pass and returns the tile table as third value
    (hand it to fps_range); results are identical either way."""
    B, H, W = ri.shape
    P = H * W
    temp = torch.empty((B, P), dtype=torch.float32, device=_dev(ri))
    info = torch.empty((B, _lib.INFO_INTS), dtype=torch.int32, device=_dev(ri))
    tab = None
    if fps_table:
        tab = torch.empty(_lib.lib().rpcc_fps_table_bytes(B, H, W) // 4, dtype=torch.float32, device=_dev(ri))
    check(_lib.lib().rpcc_ground_mask(ptr(ri), ptr(tm), ptr(ground), float(threshold), B, H, W, ptr(temp), ptr(info),
                                      ptr(tab), stream()))
    return (temp, info, tab) if fps_table else (temp, info)


def fps_xyz(points, npoint, temp=None, bruteforce=False, fma=0, cuda_tie=False):
    """a6 on an explicit point list: furthest_point_sampling_wrapper(b,n,m,points,temp,idx)
    (ops/fps/src/sampling.cpp:24-37).  points f32 [B,N,3] -> idx i32 [B,npoint].  bruteforce=True: the
    one-pass-per-centre kernel (test reference of the tile-pruned one; identical results).  fma / cuda_tie: the CUDA
    binary's contraction of sampling_gpu.cu:64 and its reduction tree's tie rule (_lib.fps_mode_flags; None = environment)."""
    B, N, _ = points.shape
    if temp is None:
        temp = torch.full((B, N), 1e10, dtype=torch.float32, device=_dev(points))
    idx = torch.empty((B, npoint), dtype=torch.int32, device=_dev(points))
    mode = _lib.fps_mode_flags(fma, cuda_tie)
    if mode:
        check(_lib.lib().rpcc_fps_xyz_mode(B, N, npoint, ptr(points), ptr(temp), ptr(idx), mode, stream()))
        return idx
    fn = _lib.lib().rpcc_fps_xyz_bruteforce if bruteforce else _lib.lib().rpcc_fps_xyz
    check(fn(B, N, npoint, ptr(points), ptr(temp), ptr(idx), stream()))
    return idx


def fps_xyz_probe(points):
    """Which kernel fps_xyz gives each list of points f32 [B,N,3]: i32 [B], -1 = one pass per centre (no locality in the point order),
    0 = tile-pruned (consecutive points are neighbours).  Test hook; the indices are the same either way."""
    B, N, _ = points.shape
    marks = torch.empty((B,), dtype=torch.int32, device=_dev(points))
    check(_lib.lib().rpcc_fps_xyz_probe(B, N, ptr(points), ptr(marks), stream()))
    return marks


def fps_range(ri, tm, temp, info, M, fps_table=None, cen_pix=None, centers=None, bruteforce=False, fma=0, cuda_tie=False):
    """a6 on the range image.  fma / cuda_tie: the CUDA-binary modes (_lib.fps_mode_flags); they need temp / info from
    ground_mask(..., fps_table=False)."""
    B, H, W = ri.shape
    cen_pix = torch.empty((B, M), dtype=torch.int32, device=_dev(ri)) if cen_pix is None else cen_pix
    centers = torch.empty((B, M, 3), dtype=torch.float32, device=_dev(ri)) if centers is None else centers
    flags = (_lib.FPS_BRUTEFORCE if bruteforce else 0) | _lib.fps_mode_flags(fma, cuda_tie)
    check(_lib.lib().rpcc_fps_range(ptr(ri), ptr(tm), ptr(temp), ptr(info), B, H, W, M, ptr(cen_pix), ptr(centers),
                                    flags, ptr(fps_table), stream()))
    return cen_pix, centers


def _stage_entry(name, M, seg=None):
    """The stage entry `name` for cluster count M: the byte-label one up to 254, its uint16 form (`name`_wide) up to 1022 -- and the label tensor's dtype
    must be the one the entry reads."""
    wide = is_wide(M)
    if wide and M > _lib.MAX_CLUSTERS_MID:
        check_cluster_num(M, stage="mid")
    if seg is not None:
        assert seg.dtype == (torch.uint16 if wide else torch.uint8), "labels of cluster_num = %d are %s on the device" % (M, "uint16" if wide else "uint8")
    return getattr(_lib.lib(), name + "_wide" if wide else name)


def label_dtype(M):
    return torch.uint16 if is_wide(M) else torch.uint8


def assign(ri, tm, ground, centers, out=None):
    """a7 -> labels u8 [B,H,W] (uint16 for 255 .. 1022 centres: rpcc_assign_wide)."""
    B, H, W = ri.shape
    M = centers.shape[1]
    seg = torch.empty((B, H, W), dtype=label_dtype(M), device=_dev(ri)) if out is None else out
    check(_stage_entry("rpcc_assign", M, seg)(ptr(ri), ptr(tm), ptr(ground), ptr(centers), B, H, W, M, ptr(seg), stream()))
    return seg


def workspace(B, P, M, device, total_points=0, general=False):
    """Work buffer of the fused entry; general=True: large enough for the plane model / the non-uniform framework too."""
    fn = _lib.lib().rpcc_wide_workspace_bytes if is_wide(M) else _lib.lib().rpcc_workspace_bytes_general if general else _lib.lib().rpcc_workspace_bytes
    n = fn(B, P, M, int(total_points))
    return torch.empty((n + 255) // 256 * 256, dtype=torch.uint8, device=device)


def point_model(ri, seg, ground, M, ws=None):
    B = ri.shape[0]
    P = ri[0].numel()
    K = M + 2
    ws = workspace(B, P, M, _dev(ri)) if ws is None else ws
    model = torch.empty((B, K, 4), dtype=torch.float32, device=_dev(ri))
    counts = torch.empty((B, K), dtype=torch.int32, device=_dev(ri))
    check(_stage_entry("rpcc_point_model", M, seg)(ptr(ri), ptr(seg), ptr(ground), B, P, M, ptr(model), ptr(counts), ptr(ws), stream()))
    return model, counts


def predict_quantize(ri, tm, seg, model, acc, M, want_pred=False, int16=False, ws=None, label_acc=None,
                     residual=None, q_out=None, nnz_out=None):
    """a10+a11(+a13).  label_acc f32 [B,K]: per-label steps (non-uniform); residual f32 [B,P]: use this
    residual instead of ri - pred.  -> (q [B,P] label-ordered, nnz [B], pred or None).  q_out / nnz_out: write into
    these buffers (entries of q past nnz are left as they are) instead of fresh zero-filled ones."""
    B = ri.shape[0]
    P = ri[0].numel()
    ws = workspace(B, P, M, _dev(ri)) if ws is None else ws
    if q_out is not None:
        assert q_out.dtype == (torch.int16 if int16 else torch.int32) and q_out.numel() == B * P
    q = torch.zeros((B, P), dtype=torch.int16 if int16 else torch.int32, device=_dev(ri)) if q_out is None else q_out
    nnz = torch.empty((B,), dtype=torch.int32, device=_dev(ri)) if nnz_out is None else nnz_out
    pred = torch.empty((B, P), dtype=torch.float32, device=_dev(ri)) if want_pred else None
    check(_stage_entry("rpcc_predict_quantize", M, seg)(ptr(ri), ptr(tm), ptr(seg), ptr(model), ptr(label_acc), ptr(residual),
                                                        float(acc), B, P, M, ptr(q) if int16 else None,
                                                        None if int16 else ptr(q), ptr(nnz), ptr(pred), ptr(ws), stream()))
    return q, nnz, pred


def intra_predict(seg, model, tm):
    """a10 alone -> pred f32 [B,H,W]."""
    B = seg.shape[0]
    P = seg[0].numel()
    pred = torch.empty(tuple(seg.shape), dtype=torch.float32, device=_dev(seg))
    fn = _lib.lib().rpcc_intra_predict_wide if seg.dtype == torch.uint16 else _lib.lib().rpcc_intra_predict    # (any label count: a row lookup per pixel)
    check(fn(ptr(seg), ptr(model), ptr(tm), B, P, model.shape[1] - 2, ptr(pred), stream()))
    return pred


def extract_features(ri, seg, feature_region=3, segments=8, sharp_num=4, less_sharp_num=8, flat_num=6):
    """a12 -> (feat f32 [B,H,W], key_point_map u8 [B,H,W])."""
    B, H, W = seg.shape
    feat = torch.empty((B, H, W), dtype=torch.float32, device=_dev(seg))
    kp = torch.empty((B, H, W), dtype=torch.uint8, device=_dev(seg))
    fn = _lib.lib().rpcc_extract_features_wide if seg.dtype == torch.uint16 else _lib.lib().rpcc_extract_features
    check(fn(ptr(ri), ptr(seg), B, H, W, feature_region, segments, sharp_num, less_sharp_num, flat_num, ptr(feat), ptr(kp), stream()))
    return feat, kp


def salience(seg, kp, level_kp_num, level_acc, ground_level, M):
    """a13 (levels) -> (salience u8 [B,K], label_acc f32 [B,K])."""
    B = seg.shape[0]
    P = seg[0].numel()
    K = M + 2
    L = len(level_kp_num)
    lk = (C.c_int32 * L)(*[int(x) for x in level_kp_num])
    la = (C.c_float * L)(*[float(x) for x in level_acc])
    sal = torch.empty((B, K), dtype=torch.uint8, device=_dev(seg))
    lacc = torch.empty((B, K), dtype=torch.float32, device=_dev(seg))
    check(_stage_entry("rpcc_salience", M, seg)(ptr(seg), ptr(kp), lk, la, L, int(ground_level), B, P, M, ptr(sal), ptr(lacc), stream()))
    return sal, lacc


def backproject(ri, tm):
    """a3: pc = ri[...,None] * transform_map -> f32 [B,H,W,3]."""
    B = ri.shape[0]
    P = ri[0].numel()
    pc = torch.empty(tuple(ri.shape[:3]) + (3,), dtype=torch.float32, device=_dev(ri))
    check(_lib.lib().rpcc_backproject(ptr(ri), ptr(tm), B, P, ptr(pc), stream()))
    return pc


def codec_workspace(B, P, M, device):
    """Work buffer of the contour codec and the decoder (cluster_num > 254: the wide decoder's sort buffers as well)."""
    n = _lib.lib().rpcc_wide_workspace_bytes(B, P, M, 0) if is_wide(M) else _lib.lib().rpcc_codec_workspace_bytes(B, P, M)
    return torch.empty(n, dtype=torch.uint8, device=device)


def contour_encode(seg, M=DEFAULT_CLUSTERS, ws=None):
    """f1: seg u8 [B,H,W] -> (contour_bits u8 [B,ceil(P/8)], idx_sequence u16 [B,P], nseq i32 [B])."""
    B, H, W = seg.shape
    P = H * W
    ws = codec_workspace(B, P, M, _dev(seg)) if ws is None else ws
    bits = torch.empty((B, (P + 7) // 8), dtype=torch.uint8, device=_dev(seg))
    seq = torch.empty((B, P), dtype=torch.uint16, device=_dev(seg))
    nseq = torch.empty((B,), dtype=torch.int32, device=_dev(seg))
    entry = _lib.lib().rpcc_contour_encode_wide if seg.dtype == torch.uint16 else _lib.lib().rpcc_contour_encode
    check(entry(ptr(seg), B, H, W, ptr(bits), ptr(seq), ptr(nseq), ptr(ws), stream()))
    return bits, seq, nseq


def contour_decode(bits, seq, H, W, M=DEFAULT_CLUSTERS, ws=None):
    """f3: recover_map on the packed contour bits -> seg u8 [B,H,W]."""
    B = bits.shape[0]
    P = H * W
    ws = codec_workspace(B, P, M, _dev(bits)) if ws is None else ws
    wide = is_wide(M)
    seg = torch.empty((B, H, W), dtype=torch.uint16 if wide else torch.uint8, device=_dev(bits))
    check((_lib.lib().rpcc_contour_decode_wide if wide else _lib.lib().rpcc_contour_decode)(ptr(bits), ptr(seq), B, H, W, ptr(seg), ptr(ws), stream()))
    return seg


def decode(seg, q16, model, tm, level_acc, salience=None, want_points=False, ws=None):
    """f3: dequantise + predict + reconstruct.  level_acc: float (uniform) or sequence (non-uniform, with
    salience u8 [B,K]).  -> (ri_rec f32 [B,H,W], pc_rec f32 [B,H,W,3] or None)."""
    B, H, W = seg.shape
    P = H * W
    M = model.shape[1] - 2
    ws = codec_workspace(B, P, M, _dev(seg)) if ws is None else ws
    uniform = salience is None
    acc = [float(level_acc)] if uniform else [float(a) for a in level_acc]
    arr = (C.c_double * len(acc))(*acc)
    rec = torch.empty((B, H, W), dtype=torch.float32, device=_dev(seg))
    pc = torch.empty((B, H, W, 3), dtype=torch.float32, device=_dev(seg)) if want_points else None
    entry = _lib.lib().rpcc_decode_wide if seg.dtype == torch.uint16 else _lib.lib().rpcc_decode
    check(entry(ptr(seg), ptr(q16), ptr(model), ptr(tm), arr, 0 if uniform else len(acc),
                ptr(salience), B, P, M, ptr(rec), ptr(pc), ptr(ws), stream()))
    return rec, pc


def pack_payload(q16, nnz, packed=None, capacity=None, total=None):
    """f2: the batch's residual stream with the frames back to back (what the container / the payload gather holds):
    packed[:sum(nnz)] = concat_b q16[b][:nnz[b]].  `packed` (i16, >= capacity entries) and `total` (i64 [1]) are
    allocated when not given; capacity defaults to B*P.  -> (packed, total)."""
    B = q16.shape[0]
    P = q16.numel() // B
    if packed is None:
        packed = torch.zeros((B * P if capacity is None else int(capacity),), dtype=torch.int16, device=_dev(q16))
    capacity = packed.numel() if capacity is None else int(capacity)
    assert packed.numel() >= capacity and packed.dtype == torch.int16
    if total is None:
        total = torch.zeros((1,), dtype=torch.int64, device=_dev(q16))
    check(_lib.lib().rpcc_pack_payload(ptr(q16), ptr(nnz), B, P, ptr(packed), capacity, ptr(total), stream()))
    return packed, total


def check_cluster_num(M, wide=True, stage=None):
    """cluster_num as this build takes it.  The reference accepts any value (cfgs/compressor.yaml:22; its labels travel as uint16,
    utils/compress_utils.py:160).  Up to 254 the device keeps a pixel's label 0 .. cluster_num + 1 in ONE byte (the tuned kernels); above,
    the batch front-end (BatchBuffers / compress_batch / contour_encode / decode) takes the uint16 entries (rpcc_*_wide: the same results by
    plain kernels) up to 65 533.  wide=False: a caller that only has the byte-label stage entries (the mirror classes' per-stage calls) --
    a larger value is refused by name, before any buffer is allocated."""
    M = int(M)
    if stage == "mid":    # the uint16 stage entries (segmentation, point model, prediction, uniform quantiser): RPCC_MAX_CLUSTERS_MID
        if not 1 <= M <= _lib.MAX_CLUSTERS_MID:
            raise _lib.RpccError("cluster_num = %d: the stage-by-stage entries for segmentation, point model, prediction and the uniform quantiser support 1 <= "
                                 "cluster_num <= %d (their label tables live in LDS: RPCC_MAX_CLUSTERS_MID in include/rpcc_hip.h); pipeline.BatchCompressor "
                                 "takes up to 65533" % (M, _lib.MAX_CLUSTERS_MID))
        return M
    top = _lib.MAX_CLUSTERS_WIDE if wide else _lib.MAX_CLUSTERS
    if not 1 <= M <= top:
        raise _lib.RpccError("cluster_num = %d: %s supports 1 <= cluster_num <= %d (%s); the reference's default is 100 (cfgs/compressor.yaml:22)"
                             % (M, "this build" if wide else "the stage-by-stage entries", top,
                                "labels are uint16 in the container" if wide else
                                "their device labels are uint8: RPCC_MAX_CLUSTERS in include/rpcc_hip.h; pipeline.BatchCompressor takes up to 65533"))
    return M


def is_wide(M):
    return int(M) > _lib.MAX_CLUSTERS


class BatchBuffers:
    """Device buffers of one batch (B frames, one geometry), allocated once and reused.  general=True adds what the
    plane model / the non-uniform framework need (salience levels, key-point map, the larger work buffer)."""

    def __init__(self, B, geom, M, device, max_points=None, general=False):
        M = check_cluster_num(M)
        self.wide = is_wide(M)     # uint16 labels, the rpcc_*_wide entries
        general = general or self.wide   # (the wide entry serves every framework / model combination from one workspace)
        P = geom.H * geom.W
        K = M + 2
        self.B, self.P, self.M, self.K, self.geom = B, P, M, K, geom
        f32, i32 = torch.float32, torch.int32
        self.ri = torch.empty((B, geom.H, geom.W), dtype=f32, device=device)
        self.seg = torch.empty((B, geom.H, geom.W), dtype=torch.uint16 if self.wide else torch.uint8, device=device)
        self.cen_pix = torch.empty((B, M), dtype=i32, device=device)
        self.centers = torch.empty((B, M, 3), dtype=f32, device=device)
        self.model = torch.empty((B, K, 4), dtype=f32, device=device)
        self.counts = torch.empty((B, K), dtype=i32, device=device)
        self.q16 = torch.empty((B, P), dtype=torch.int16, device=device)
        self.nnz = torch.empty((B,), dtype=i32, device=device)
        self.info = torch.empty((B, _lib.INFO_INTS), dtype=i32, device=device)
        # max_points: initial capacity (sum of N over the batch) of the projection's record space -- about 40 B of workspace per point
        # and 48 KB per frame (INTEGRATION.md "Memory").  Default one point per pixel: compress_batch re-allocates the workspace
        # when a batch holds more (dense or dual-return sweeps), so the default no longer reserves twice that up front.
        self.max_points = int(max_points) if max_points is not None else B * P
        self.general = bool(general)
        self.ws = workspace(B, P, M, device, self.max_points, general=self.general)
        self.salience = torch.zeros((B, K), dtype=torch.uint8, device=device) if general else None
        self.key_point_map = torch.empty((B, geom.H, geom.W), dtype=torch.uint8, device=device) if general else None


class FpsTimer:
    """rpcc_timer handle (bench.py): pass as compress_batch(timer=...) to time that call's FPS launch with HIP events."""

    def __init__(self):
        self.h = C.c_void_p(_lib.lib().rpcc_timer_create())
        assert self.h.value, "rpcc_timer_create failed"

    def reserve(self, launches):
        """Create the events of the next `launches` timed launches now (outside the caller's timed region)."""
        check(_lib.lib().rpcc_timer_reserve(self.h, int(launches)))

    def read(self):
        """-> (accumulated ms, launches) since the last read; synchronises the recorded events."""
        ms, n = C.c_double(0), C.c_int(0)
        check(_lib.lib().rpcc_timer_read(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def __del__(self):
        try:
            _lib.lib().rpcc_timer_destroy(self.h)
        except Exception:
            pass


def nonuniform_cfg(acc, cfg=None):
    """QuantizationModule's non-uniform settings (utils/compress_utils.py:36-54) as the C struct: acc = base step
    (2 * accuracy), cfg = compressor.yaml keys (level_key_point_num, level_delta_acc, ground_salience_level,
    feature_region, segments, sharp_num, less_sharp_num, flat_num)."""
    c = cfg or {}
    lk = list(c.get("level_key_point_num", (30, 10, 3, 0)))
    la = (np.array([acc] * len(lk)) + np.array(c.get("level_delta_acc", (0, 0.02, 0.04, 0.06)))).astype(np.float32)
    assert 1 <= len(lk) <= 8
    nu = _lib.NonuniformCfg()
    nu.levels = len(lk)
    for i, v in enumerate(lk):
        nu.level_kp_num[i] = int(v)
        nu.level_acc[i] = float(la[i])
    nu.ground_level = int(c.get("ground_salience_level", 2))
    nu.feature_region, nu.segments = int(c.get("feature_region", 3)), int(c.get("segments", 8))
    nu.sharp_num, nu.less_sharp_num, nu.flat_num = int(c.get("sharp_num", 4)), int(c.get("less_sharp_num", 8)), int(c.get("flat_num", 6))
    return nu


def compress_batch(xyz, offsets, tm, ground, buf, ground_threshold=0.1, acc=0.04, ground_seed=-1, frame_ids=None,
                   fps_bruteforce=False, timer=None, model_method="point", angle_threshold=75, plane_seed=0, nonuniform=None,
                   fps_fma=None, fps_cuda_tie=None, project_flags=0):
    """Fused a2..a13 for a batch, one call: FPS segmentation, point or plane model, uniform or non-uniform framework
    (tools/compress.py:93-125).  xyz: f32 [total,3], or the stored (x, y, z, intensity) rows f32 [total,4] (16-byte stride,
    no host-side slice).  ground f64 [B,4]: injected models when ground_seed < 0, otherwise output of the seeded
    ground RANSAC run inside the call (frame b draws with ground_seed + frame_ids[b]; frame_ids: stable identities,
    e.g. utils.frame_identity(path); default the batch position).  model_method "plane": rpcc_plane_model's seeded fits
    (plane_seed, frame_ids) with the reference's angle validation.  nonuniform: a nonuniform_cfg() struct -> key points,
    salience levels (buf.salience) and per-label steps; None = uniform framework with step `acc`.
    fps_fma / fps_cuda_tie: the CUDA-binary FPS modes (_lib.fps_mode_flags); None = the environment variables RPCC_FPS_FMA
    (0 / 1 / 2) and RPCC_FPS_TIE_CUDA, so every front-end (tools, pipeline, loader) honours them.
    project_flags: 0, PROJECT_ORDER_PROBE (frames in scanner order take the record-free projection kernel, chosen by a probe),
    PROJECT_FORCE_ORDERED (see project())."""
    io = _batch_io(xyz, offsets, tm, ground, buf, ground_seed, frame_ids, fps_bruteforce, timer, model_method, angle_threshold, plane_seed,
                   nonuniform, fps_fma, fps_cuda_tie, project_flags)
    entry = _lib.lib().rpcc_compress_batch_wide if buf.wide else _lib.lib().rpcc_compress_batch   # (cluster_num > 254: uint16 labels)
    check(entry(C.byref(io), buf.B, buf.geom, buf.M, float(ground_threshold), float(acc), ptr(buf.ws), stream()))
    return buf


STAGE_PROJECT, STAGE_GROUND, STAGE_MASK, STAGE_FPS, STAGE_LABELS, STAGE_PLANES, STAGE_QUANTISE = (1 << i for i in range(7))


def compress_batch_stages(stage_mask, xyz, offsets, tm, ground, buf, ground_threshold=0.1, acc=0.04, **kw):
    """The stages of compress_batch whose bits are set in stage_mask (STAGE_*), in order, on the current stream: for a caller that interleaves the
    stages of several batches on several streams (tools_dev/tick_bench.py).  Every stage needs its predecessors to have run on the same buffers."""
    assert not buf.wide
    io = _batch_io(xyz, offsets, tm, ground, buf, **kw)
    check(_lib.lib().rpcc_compress_batch_stages(C.byref(io), buf.B, buf.geom, buf.M, float(ground_threshold), float(acc), ptr(buf.ws), int(stage_mask), stream()))
    return buf


def _batch_io(xyz, offsets, tm, ground, buf, ground_seed=-1, frame_ids=None, fps_bruteforce=False, timer=None, model_method="point",
              angle_threshold=75, plane_seed=0, nonuniform=None, fps_fma=None, fps_cuda_tie=None, project_flags=0):
    """The rpcc_batch_io of one geometry group (compress_batch's arguments); grows the group's workspace when the batch holds more points."""
    general = model_method != "point" or nonuniform is not None
    assert not general or buf.general, "BatchBuffers(..., general=True) is needed for the plane model / non-uniform framework"
    if xyz.shape[0] > buf.max_points:
        buf.max_points = int(xyz.shape[0])
        buf.ws = workspace(buf.B, buf.P, buf.M, xyz.device, buf.max_points, general=buf.general)
    fid = _frame_ids(frame_ids, buf.B, xyz.device)
    buf._frame_ids = fid          # keep the tensor alive until the stream has consumed it
    buf._nonuniform = nonuniform  # (host struct read during the call only; kept for symmetry)
    return BatchIO(ptr(xyz).value, ptr(offsets).value, int(xyz.shape[0]), ptr(tm).value, ptr(ground).value,
                   int(ground_seed), ptr(fid).value if fid is not None else None, ptr(buf.ri).value, ptr(buf.seg).value,
                   ptr(buf.cen_pix).value, ptr(buf.centers).value, ptr(buf.model).value, ptr(buf.counts).value,
                   ptr(buf.q16).value, ptr(buf.nnz).value, ptr(buf.info).value,
                   (_lib.FPS_BRUTEFORCE if fps_bruteforce else 0) | _lib.fps_mode_flags(fps_fma, fps_cuda_tie) | _project_flags(project_flags),
                   timer.h if timer is not None else None,
                   0 if model_method == "point" else 1, angle_cos_cut(angle_threshold) if model_method != "point" else 0.0,
                   int(plane_seed), C.addressof(nonuniform) if nonuniform is not None else None,
                   ptr(buf.salience).value if nonuniform is not None else None,
                   ptr(buf.key_point_map).value if nonuniform is not None else None,
                   _point_stride(xyz))


def compress_batch_mixed(groups, ground_threshold=0.1, acc=0.04):
    """Fused a2..a13 for a batch that holds sweeps of several lidar geometries (variable H x W: BASELINE configs[4]), one call on the
    current stream.  groups: one dict per geometry with compress_batch's arguments (xyz, offsets, tm, ground, buf and, optionally,
    ground_seed, frame_ids, model_method, angle_threshold, plane_seed, nonuniform, fps_bruteforce, fps_fma, fps_cuda_tie); every buf is
    a BatchBuffers of that geometry with the same cluster count.  The kernels with one workgroup per frame or label (ground RANSAC,
    FPS, plane fits) run once over all groups (include/rpcc_hip.h: rpcc_compress_batch_mixed).  -> the list of the groups' buffers."""
    G = len(groups)
    assert 1 <= G <= _lib.MAX_GROUPS, "at most %d geometry groups per call" % _lib.MAX_GROUPS
    M = groups[0]["buf"].M
    assert all(g["buf"].M == M for g in groups), "one cluster count per call"
    if is_wide(M):   # (no mixed-geometry form of the uint16-label entry: the groups one after the other, same results)
        for g in groups:
            compress_batch(ground_threshold=ground_threshold, acc=acc, **g)
        return [g["buf"] for g in groups]
    ios = (BatchIO * G)(*[_batch_io(**g) for g in groups])
    Bs = (C.c_int * G)(*[g["buf"].B for g in groups])
    geoms = (_lib.Geom * G)(*[g["buf"].geom for g in groups])
    wss = (C.c_void_p * G)(*[ptr(g["buf"].ws).value for g in groups])
    check(_lib.lib().rpcc_compress_batch_mixed(ios, Bs, geoms, G, M, float(ground_threshold), float(acc), wss, stream()))
    return [g["buf"] for g in groups]


@functools.lru_cache(maxsize=64)
def angle_cos_cut(angle_threshold_deg):
    """Largest double v with arccos(v) > pi*(angle/180): the reference rejects a plane when
    alpha.max() > threshold (utils/segment_utils.py:89); the kernel compares v <= cos_cut instead of
    evaluating arccos per pixel.  Found by bisection on numpy's own arccos."""
    thr = np.pi * (angle_threshold_deg / 180)
    lo, hi = 0.0, 1.0          # arccos(lo) > thr >= arccos(hi)
    if not np.arccos(lo) > thr:
        return -1.0            # nothing in [0,1] is rejected
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        if mid == lo or mid == hi:
            break
        if np.arccos(mid) > thr:
            lo = mid
        else:
            hi = mid
    return float(lo)


def plane_model(ri, tm, seg, M, angle_threshold=75, seed=0, ground=None, want_counts=False, frame_ids=None, inject=None):
    """a9 -> model f32 [B,K,4] (and counts i32 [B,K]).  Label k of frame b draws with hash(seed, frame_ids[b], k).
    inject f64 [B,K,4] (test hook): planes used instead of the RANSAC results."""
    B = ri.shape[0]
    P = ri[0].numel()
    K = M + 2
    ws = torch.empty(_lib.lib().rpcc_plane_workspace_bytes(B, P, M), dtype=torch.uint8, device=_dev(ri))
    model = torch.empty((B, K, 4), dtype=torch.float32, device=_dev(ri))
    counts = torch.empty((B, K), dtype=torch.int32, device=_dev(ri))
    fid = _frame_ids(frame_ids, B, _dev(ri))
    check(_stage_entry("rpcc_plane_model", M, seg)(ptr(ri), ptr(tm), ptr(seg), ptr(ground), B, P, M, angle_cos_cut(angle_threshold),
                                                   int(seed) & 0xFFFFFFFF, ptr(fid), ptr(inject), ptr(model), ptr(counts), ptr(ws), stream()))
    return (model, counts) if want_counts else model


def compress_batch_general(xyz, offsets, tm, ground, buf, cc, fit_ground, frame_ids=None):
    """Stage-by-stage batch path for the configurations the fused entry does not cover (non-uniform
    framework and / or plane model).  `cc` is a pipeline.BatchCompressor (settings holder).  Fills `buf`
    like compress_batch and returns the salience levels u8 [B,K] (None for the uniform framework)."""
    B, M, geom = buf.B, buf.M, buf.geom
    project(xyz, offsets, geom, ri=buf.ri)
    if fit_ground:
        g, _ = ground_ransac(buf.ri, tm, seed=cc.seed, frame_ids=frame_ids)
        ground.copy_(g)
    if _lib.fps_mode_flags(None, None):   # RPCC_FPS_FMA / RPCC_FPS_TIE_CUDA in the environment
        temp, info = ground_mask(buf.ri, tm, ground, cc.ground_threshold, fps_table=False)
        buf.info.copy_(info)
        _, centers = fps_range(buf.ri, tm, temp, info, M, cen_pix=buf.cen_pix, centers=buf.centers, fma=None, cuda_tie=None)
    else:
        temp, info, tab = ground_mask(buf.ri, tm, ground, cc.ground_threshold, fps_table=True)
        buf.info.copy_(info)
        _, centers = fps_range(buf.ri, tm, temp, info, M, fps_table=tab, cen_pix=buf.cen_pix, centers=buf.centers)
    assign(buf.ri, tm, ground, centers, out=buf.seg)
    if cc.model_method == "point":
        model, counts = point_model(buf.ri, buf.seg, ground, M, ws=buf.ws)
    else:
        model, counts = plane_model(buf.ri, tm, buf.seg, M, angle_threshold=cc.cfg.get("plane_angle_threshold", 75),
                                    seed=cc.seed, ground=ground, want_counts=True, frame_ids=frame_ids)
    buf.model.copy_(model)
    buf.counts.copy_(counts)
    sal = label_acc = None
    if not cc.uniform:
        c = cc.cfg
        _, kp = extract_features(buf.ri, buf.seg, c.get("feature_region", 3), c.get("segments", 8), c.get("sharp_num", 4),
                                 c.get("less_sharp_num", 8), c.get("flat_num", 6))
        lk = c.get("level_key_point_num", (30, 10, 3, 0))
        la = np.array([cc.acc] * len(lk)) + np.array(c.get("level_delta_acc", (0, 0.02, 0.04, 0.06)))
        sal, label_acc = salience(buf.seg, kp, lk, la.astype(np.float32), c.get("ground_salience_level", 2), M)
    predict_quantize(buf.ri, tm, buf.seg, buf.model, cc.acc, M, int16=True, ws=buf.ws, label_acc=label_acc,
                     q_out=buf.q16, nnz_out=buf.nnz)
    return sal

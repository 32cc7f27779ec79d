"""Mirror of the reference's utils/compress_utils.py: QuantizationModule on the HIP path, payload
packing, the .rpcc container and the entropy back-ends (stdlib calls, excluded from the measured path).
"""
import bz2
import copy
import gzip
import os
import struct

import numpy as np
import torch

from . import ops
from .contour_utils import ContourExtractor
from .utils import load_yaml


def _dev(a, device, dtype=None):
    a = np.ascontiguousarray(a)
    if dtype is not None:
        a = a.astype(dtype)
    return torch.from_numpy(a).to(device)


class QuantizationModule:
    """utils/compress_utils.py:35-132."""

    def __init__(self, base_accuracy, level_kp_num=(30, 10, 3, 0), level_dacc=(0, 0.02, 0.04, 0.06),
                 ground_salience_level=2, feature_region=3, segments=8, sharp_num=4, less_sharp_num=8, flat_num=6,
                 uniform=True, device="cuda:0"):
        self.uniform = uniform
        self.device = torch.device(device)
        if uniform:
            self.acc = base_accuracy
        else:
            self.level_kp_num = np.array(level_kp_num)
            self.acc = np.array([base_accuracy] * len(self.level_kp_num)) + np.array(level_dacc)
            self.ground_level = ground_salience_level
            self.feature_region = feature_region
            self.segments = segments
            self.sharp_num = sharp_num
            self.less_sharp_num = less_sharp_num
            self.flat_num = flat_num

    def quantize_residual(self, residual, seg_idx, point_cloud=None, range_image=None):
        """-> (residual_quantized int32 [nnz], salience_level int32 [max+1] or None, key_point_map or None)."""
        h, w = seg_idx.shape[:2]
        M = max(int(seg_idx.max()) - 1, 1)
        K = M + 2
        # the stage entries exist for uint16 labels too (rpcc_predict_quantize_wide, rpcc_extract_features_wide, rpcc_salience_wide: cluster_num <= 1022;
        # the batch front-end, pipeline.BatchCompressor, goes up to 65533)
        ops.check_cluster_num(M, stage="mid")
        seg = _dev(seg_idx, self.device, np.uint16 if ops.is_wide(M) else np.uint8).reshape(1, h, w)
        res = _dev(residual, self.device, np.float32).reshape(1, h * w)
        dummy_model = torch.zeros((1, K, 4), dtype=torch.float32, device=self.device)
        tm = torch.zeros((h * w, 3), dtype=torch.float32, device=self.device)
        ri0 = torch.zeros((1, h, w), dtype=torch.float32, device=self.device)
        if self.uniform:
            q, nnz, _ = ops.predict_quantize(ri0, tm, seg, dummy_model, self.acc, M, residual=res)
            return q[0, : int(nnz[0])].cpu().numpy(), None, None
        ri = _dev(range_image, self.device, np.float32).reshape(1, h, w)
        _, kp = ops.extract_features(ri, seg, self.feature_region, self.segments, self.sharp_num, self.less_sharp_num,
                                     self.flat_num)
        sal, label_acc = ops.salience(seg, kp, self.level_kp_num, self.acc.astype(np.float32), self.ground_level, M)
        q, nnz, _ = ops.predict_quantize(ri0, tm, seg, dummy_model, 0.0, M, residual=res, label_acc=label_acc)
        nrow = int(seg_idx.max()) + 1
        return (q[0, : int(nnz[0])].cpu().numpy(), sal[0, :nrow].cpu().numpy().astype(np.int32),
                kp[0].cpu().numpy().astype(np.int32))

    def dequantize_residual(self, quantized_residual, seg_idx, salience_level=None):
        """utils/compress_utils.py:114-132 -> f32 [H,W,1] (decoder; on the device via rpcc_decode with a
        zero model, so pred = 0 and rec = residual)."""
        h, w = seg_idx.shape[:2]
        seg = _dev(seg_idx, self.device, np.uint8).reshape(1, h, w)
        M = max(int(seg_idx.max()) - 1, 1)
        K = M + 2
        q = torch.zeros((1, h * w), dtype=torch.int16, device=self.device)
        qq = _dev(quantized_residual, self.device, np.int16)
        q[0, : qq.numel()] = qq
        model = torch.zeros((1, K, 4), dtype=torch.float32, device=self.device)
        tm = torch.zeros((h * w, 3), dtype=torch.float32, device=self.device)
        if self.uniform:
            rec, _ = ops.decode(seg, q, model, tm, self.acc)
        else:
            sal = torch.zeros((1, K), dtype=torch.uint8, device=self.device)
            s = _dev(salience_level, self.device, np.uint8)
            sal[0, : s.numel()] = s
            rec, _ = ops.decode(seg, q, model, tm, list(self.acc), salience=sal)
        return np.expand_dims(rec[0].cpu().numpy(), -1)


def compress_point_cloud(basic_compressor, plane_param, cluster_idx, salience_level, nonzero_residual_quantized,
                         ground_residual_quantized=None, cluster_residual_quantized=None, point_cloud=None,
                         range_image=None, full=False):
    """utils/compress_utils.py:138-164: casts + contour + packbits + per-array entropy coding."""
    original_data = {"residual_quantized": np.asarray(nonzero_residual_quantized).astype(np.int16)}
    if full:
        if point_cloud is not None:
            original_data["point_cloud"] = point_cloud.astype(np.float32)
        if range_image is not None:
            original_data["range_image"] = range_image.astype(np.float32)
        if ground_residual_quantized is not None:
            original_data["ground_residual"] = ground_residual_quantized.astype(np.int16)
        if cluster_residual_quantized is not None:
            original_data["cluster_residual"] = cluster_residual_quantized.astype(np.int16)
    if salience_level is not None:
        original_data["salience_level"] = np.asarray(salience_level).astype(np.uint8)
    contour_map, idx_sequence = ContourExtractor.extract_contour(cluster_idx)
    original_data["contour_map"] = np.packbits(contour_map.astype(bool), axis=None).astype(np.uint8)
    original_data["idx_sequence"] = idx_sequence.astype(np.uint16)
    original_data["plane_param"] = np.asarray(plane_param).astype(np.float32)
    return original_data, basic_compressor.compress_dict(original_data)


_ORDER = ("contour_map", "idx_sequence", "plane_param", "residual_quantized")


def pack_bitstream(compressed_data, uniform=True):
    """The .rpcc container (utils/compress_utils.py:167-179): [int32 length | bytes] per array."""
    keys = (() if uniform else ("salience_level",)) + _ORDER
    return b"".join(struct.pack("i", len(compressed_data[k])) + bytes(compressed_data[k]) for k in keys)


def pack_frames(basic_compressor, frames, uniform=True):
    """The .rpcc byte strings of several frames: compress_dict + pack_bitstream per frame, with the bzip2 back-end done by
    ONE call into librpcc_host.so for the whole list (include/rpcc_host.h) -- a pool thread then enters the interpreter once
    per chunk of frames instead of once per array.  Same bytes as the per-frame path (tests/test_host_pack.py)."""
    from . import _lib
    host = None
    if basic_compressor.method_name == "bzip2" and frames:
        try:
            host = _lib.host_lib()     # not built / not loadable / stale: remembered by _lib, the Python path below gives the same bytes
        except _lib.RpccError:
            host = None
    if host is None:
        return [pack_bitstream(basic_compressor.compress_dict(od), uniform=uniform) for od in frames]
    keys = (() if uniform else ("salience_level",)) + _ORDER
    arrs = [np.ascontiguousarray(od[k]) for od in frames for k in keys]
    n, na = len(frames), len(keys)
    ptrs = np.fromiter((a.ctypes.data for a in arrs), dtype=np.uint64, count=n * na)
    lens = np.fromiter((a.nbytes for a in arrs), dtype=np.uint32, count=n * na)
    # bzip2 never expands by more than 1 % + 600 bytes; every frame gets the room its own arrays need
    room = (lens.reshape(n, na).astype(np.int64) * 101 // 100 + 604).sum(1)
    offs = np.zeros(n + 1, np.uint64)
    offs[1:] = np.cumsum(room)
    out = np.empty(int(offs[-1]), np.uint8)
    out_len = np.zeros(n, np.uint32)
    rc = host.rpcc_host_pack_bz2(n, na, ptrs.ctypes.data, lens.ctypes.data, out.ctypes.data, offs.ctypes.data, out_len.ctypes.data)
    if rc == -1:
        raise RuntimeError("rpcc_host_pack_bz2: bad argument")
    if rc != 0:
        raise RuntimeError("rpcc_host_pack_bz2 failed on frame %d of the chunk" % (-rc - 16))
    return [out[int(offs[i]): int(offs[i]) + int(out_len[i])].tobytes() for i in range(n)]


def unpack_bitstream(blob, uniform=True):
    out, off = {}, 0
    for k in (() if uniform else ("salience_level",)) + _ORDER:
        # (the reference reads whatever f.read(n) returns; a truncated or foreign file then fails somewhere inside the entropy decoder)
        if off + 4 > len(blob):
            raise ValueError("bitstream ends before the length of '%s' (%d bytes; written with the %s framework?)"
                             % (k, len(blob), "non-uniform" if uniform else "uniform"))
        (n,) = struct.unpack_from("i", blob, off)
        if n < 0 or off + 4 + n > len(blob):
            raise ValueError("bitstream: payload '%s' claims %d bytes, %d are left" % (k, n, len(blob) - off - 4))
        out[k] = blob[off + 4: off + 4 + n]
        off += 4 + n
    return out


def save_compressed_bitstream(file, compressed_data, uniform=True):
    with open(file, "wb") as f:
        f.write(pack_bitstream(compressed_data, uniform))


def read_compressed_bitstream(file, uniform=True):
    with open(file, "rb") as f:
        return unpack_bitstream(f.read(), uniform)


def decompress_point_cloud(compressed_data, basic_compressor, model_num, H, W):
    """utils/compress_utils.py:199-214 -> (residual_quantized int16, idx_map, salience_level, plane_param)."""
    d = basic_compressor.decompress_dict(compressed_data)
    plane_param = np.ndarray(shape=(model_num, 4), dtype=np.float32, buffer=d["plane_param"])
    contour_map = np.unpackbits(np.ndarray(shape=(-1,), dtype=np.uint8, buffer=d["contour_map"]))[: H * W].reshape(H, W)
    idx_sequence = np.ndarray(shape=(-1,), dtype=np.uint16, buffer=d["idx_sequence"])
    idx_map = ContourExtractor.recover_map(contour_map, idx_sequence)
    salience = np.ndarray(shape=(-1,), dtype=np.uint8, buffer=d["salience_level"]) if "salience_level" in d else None
    residual_quantized = np.ndarray(shape=(-1,), dtype=np.int16, buffer=d["residual_quantized"])
    return residual_quantized, idx_map, salience, plane_param


class BasicCompressor:
    """utils/compress_utils.py:232-310.  bzip2 / deflate are stdlib; lz4 needs the lz4 package
    (lz4==0.7.0 API in the reference), which this image does not ship."""

    METHODS = ["lz4", "bzip2", "gzip", "deflate"]

    def __init__(self, compressor_yaml=None, method_name=None):
        self.method_name = None
        if compressor_yaml is not None:
            self.method_name = load_yaml(compressor_yaml)["basic_compressor"]
        if method_name is not None:
            self.method_name = method_name
        if self.method_name is not None:
            assert self.method_name in self.METHODS, "Compression method is not existed. (lz4, bzip2, gzip, deflate)"

    def set_method(self, method_name):
        assert method_name in self.METHODS, "Compression method is not existed. (lz4, bzip2, gzip, deflate)"
        self.method_name = method_name

    def compress_dict(self, data_dict):
        return {k: self.compress(v) for k, v in data_dict.items()}

    def decompress_dict(self, data_dict):
        return {k: self.decompress(v) for k, v in data_dict.items()}

    def compress(self, np_array):
        buf = np.ascontiguousarray(np_array)
        if self.method_name == "bzip2":
            return bz2.compress(buf)
        if self.method_name in ("gzip", "deflate"):
            return gzip.compress(buf)
        if self.method_name == "lz4":
            return self._lz4().dumps(buf.tobytes())
        raise ValueError("no compression method set")

    def decompress(self, bitstream):
        if self.method_name == "bzip2":
            return bz2.decompress(bitstream)
        if self.method_name in ("gzip", "deflate"):
            return gzip.decompress(bitstream)
        if self.method_name == "lz4":
            return self._lz4().loads(bitstream)
        raise ValueError("no compression method set")

    def calc_compressed_bytes(self, np_array):
        return len(self.compress(np_array))

    @staticmethod
    def _lz4():
        try:
            import lz4
            return lz4
        except ImportError as e:
            raise RuntimeError("basic_compressor 'lz4' needs the lz4 package, which is not installed") from e

"""ctypes binding of librpcc_hip.so (include/rpcc_hip.h).  There is no CPU fallback: if the HIP
library is missing or a call fails, this raises."""
import ctypes as C
import os

import torch  # imported first so the library binds to the HIP runtime torch already loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
# RPCC_HIP_LIB: developer knob -- another build of the same ABI (an A/B of kernel variants inside one GPU session, tools_dev/ab.sh)
LIB_PATH = os.environ.get("RPCC_HIP_LIB") or os.path.join(_HERE, "lib", "librpcc_hip.so")


MAX_GROUPS = 4   # RPCC_MAX_GROUPS (include/rpcc_hip.h)


class Geom(C.Structure):
    _fields_ = [("H", C.c_int32), ("W", C.c_int32), ("horizontal_fov", C.c_float), ("vertical_max", C.c_float),
                ("vertical_min", C.c_float)]


class BatchIO(C.Structure):
    _fields_ = [("xyz", C.c_void_p), ("offsets", C.c_void_p), ("total", C.c_int64), ("tm", C.c_void_p),
                ("ground", C.c_void_p), ("ground_seed", C.c_int64), ("frame_ids", C.c_void_p), ("ri", C.c_void_p),
                ("seg", C.c_void_p), ("cen_pix", C.c_void_p), ("centers", C.c_void_p), ("model", C.c_void_p),
                ("counts", C.c_void_p), ("q16", C.c_void_p), ("nnz", C.c_void_p), ("info", C.c_void_p),
                ("flags", C.c_int32), ("timer", C.c_void_p), ("model_method", C.c_int32), ("plane_cos_cut", C.c_double),
                ("plane_seed", C.c_int64), ("nonuniform", C.c_void_p), ("salience", C.c_void_p), ("key_point_map", C.c_void_p),
                ("point_stride_bytes", C.c_int32)]


class NonuniformCfg(C.Structure):
    _fields_ = [("levels", C.c_int32), ("level_kp_num", C.c_int32 * 8), ("level_acc", C.c_float * 8), ("ground_level", C.c_int32),
                ("feature_region", C.c_int32), ("segments", C.c_int32), ("sharp_num", C.c_int32), ("less_sharp_num", C.c_int32),
                ("flat_num", C.c_int32)]


INFO_INTS = 8          # RPCC_INFO_INTS
FPS_BRUTEFORCE = 1     # RPCC_FPS_BRUTEFORCE
FPS_FMA1, FPS_FMA2, FPS_TIE_CUDA = 2, 4, 8   # RPCC_FPS_FMA1 / RPCC_FPS_FMA2 / RPCC_FPS_TIE_CUDA
MAX_CLUSTERS = 254     # RPCC_MAX_CLUSTERS: labels 0 .. cluster_num + 1 are stored as uint8 on the device
MAX_CLUSTERS_WIDE = 65533   # RPCC_MAX_CLUSTERS_WIDE: the uint16-label entries (rpcc_*_wide)
MAX_CLUSTERS_MID = 1022     # RPCC_MAX_CLUSTERS_MID: the tuned kernels on uint16 labels; the uint16 STAGE entries (rpcc_assign_wide ...)
ABI_VERSION = 103      # RPCC_ABI_VERSION: the layout of rpcc_batch_io / rpcc_geom this binding was written for


def fps_mode_flags(fma=0, cuda_tie=False):
    """The CUDA-binary FPS modes as flag bits: fma 0 / 1 / 2 (sampling_gpu.cu:64 un-fused / either nvcc contraction),
    cuda_tie (the reduction tree's winner among equal values).  None -> the environment (RPCC_FPS_FMA, RPCC_FPS_TIE_CUDA)."""
    if fma is None:
        fma = int(os.environ.get("RPCC_FPS_FMA", "0"))
    if cuda_tie is None:
        cuda_tie = os.environ.get("RPCC_FPS_TIE_CUDA", "0") not in ("", "0")
    assert fma in (0, 1, 2)
    return (FPS_FMA1 if fma == 1 else FPS_FMA2 if fma == 2 else 0) | (FPS_TIE_CUDA if cuda_tie else 0)


class RpccError(RuntimeError):
    pass


_lib = None

_VP, _I, _I64, _F, _D = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double
_SIGS = {
    "rpcc_version": (C.c_int, []),
    "rpcc_last_error": (C.c_char_p, []),
    "rpcc_project_scratch_bytes": (C.c_size_t, [_I64, _I, _I]),
    "rpcc_project_fastpath_check": (C.c_int, [_VP, _I64, Geom, _VP, _VP]),
    "rpcc_project": (C.c_int, [_VP, _VP, _I64, _I, Geom, _VP, _VP, C.c_size_t, _VP]),
    "rpcc_project_strided": (C.c_int, [_VP, _I, _VP, _I64, _I, Geom, _VP, _VP, C.c_size_t, _VP]),
    "rpcc_project_ordered": (C.c_int, [_VP, _I, _VP, _I64, _I, Geom, _VP, _VP, C.c_size_t, _I, _VP, _VP]),
    "rpcc_ground_ransac": (C.c_int, [_VP, _VP, _I, _I, C.c_uint32, _VP, _VP, _VP, _VP]),
    "rpcc_ground_mask": (C.c_int, [_VP, _VP, _VP, _D, _I, _I, _I, _VP, _VP, _VP, _VP]),
    "rpcc_fps_table_bytes": (C.c_size_t, [_I, _I, _I]),
    "rpcc_fps_xyz": (C.c_int, [_I, _I, _I, _VP, _VP, _VP, _VP]),
    "rpcc_fps_xyz_bruteforce": (C.c_int, [_I, _I, _I, _VP, _VP, _VP, _VP]),
    "rpcc_fps_xyz_mode": (C.c_int, [_I, _I, _I, _VP, _VP, _VP, _I, _VP]),
    "rpcc_fps_xyz_probe": (C.c_int, [_I, _I, _VP, _VP, _VP]),
    "rpcc_fps_range": (C.c_int, [_VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP, _VP, _I, _VP, _VP]),
    "rpcc_assign": (C.c_int, [_VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP, _VP]),
    "rpcc_point_model": (C.c_int, [_VP, _VP, _VP, _I, _I, _I, _VP, _VP, _VP, _VP]),
    "rpcc_predict_quantize": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _F, _I, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP]),
    "rpcc_intra_predict": (C.c_int, [_VP, _VP, _VP, _I, _I, _I, _VP, _VP]),
    "rpcc_assign_wide": (C.c_int, [_VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP, _VP]),
    "rpcc_point_model_wide": (C.c_int, [_VP, _VP, _VP, _I, _I, _I, _VP, _VP, _VP, _VP]),
    "rpcc_predict_quantize_wide": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _F, _I, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP]),
    "rpcc_intra_predict_wide": (C.c_int, [_VP, _VP, _VP, _I, _I, _I, _VP, _VP]),
    "rpcc_extract_features_wide": (C.c_int, [_VP, _VP, _I, _I, _I, _I, _I, _I, _I, _I, _VP, _VP, _VP]),
    "rpcc_salience_wide": (C.c_int, [_VP, _VP, C.POINTER(C.c_int32), C.POINTER(C.c_float), _I, _I, _I, _I, _I, _VP, _VP, _VP]),
    "rpcc_plane_model_wide": (C.c_int, [_VP, _VP, _VP, _VP, _I, _I, _I, _D, C.c_uint32, _VP, _VP, _VP, _VP, _VP, _VP]),
    "rpcc_extract_features": (C.c_int, [_VP, _VP, _I, _I, _I, _I, _I, _I, _I, _I, _VP, _VP, _VP]),
    "rpcc_salience": (C.c_int, [_VP, _VP, C.POINTER(C.c_int32), C.POINTER(C.c_float), _I, _I, _I, _I, _I, _VP, _VP, _VP]),
    "rpcc_backproject": (C.c_int, [_VP, _VP, _I, _I, _VP, _VP]),
    "rpcc_codec_workspace_bytes": (C.c_size_t, [_I, _I, _I]),
    "rpcc_contour_encode": (C.c_int, [_VP, _I, _I, _I, _VP, _VP, _VP, _VP, _VP]),
    "rpcc_contour_decode": (C.c_int, [_VP, _VP, _I, _I, _I, _VP, _VP, _VP]),
    "rpcc_decode": (C.c_int, [_VP, _VP, _VP, _VP, C.POINTER(C.c_double), _I, _VP, _I, _I, _I, _VP, _VP, _VP, _VP]),
    "rpcc_pack_payload": (C.c_int, [_VP, _VP, _I, _I, _VP, _I64, _VP, _VP]),
    "rpcc_plane_workspace_bytes": (C.c_size_t, [_I, _I, _I]),
    "rpcc_plane_model": (C.c_int, [_VP, _VP, _VP, _VP, _I, _I, _I, _D, C.c_uint32, _VP, _VP, _VP, _VP, _VP, _VP]),
    "rpcc_workspace_bytes": (C.c_size_t, [_I, _I, _I, _I64]),
    "rpcc_workspace_bytes_general": (C.c_size_t, [_I, _I, _I, _I64]),
    "rpcc_compress_batch": (C.c_int, [C.POINTER(BatchIO), _I, Geom, _I, _D, _F, _VP, _VP]),
    "rpcc_compress_batch_stages": (C.c_int, [C.POINTER(BatchIO), _I, Geom, _I, _D, _F, _VP, _I, _VP]),
    "rpcc_compress_batch_mixed": (C.c_int, [C.POINTER(BatchIO), C.POINTER(C.c_int), C.POINTER(Geom), _I, _I, _D, _F, C.POINTER(C.c_void_p), _VP]),
    "rpcc_wide_workspace_bytes": (C.c_size_t, [_I, _I, _I, _I64]),
    "rpcc_compress_batch_wide": (C.c_int, [C.POINTER(BatchIO), _I, Geom, _I, _D, _F, _VP, _VP]),
    "rpcc_contour_encode_wide": (C.c_int, [_VP, _I, _I, _I, _VP, _VP, _VP, _VP, _VP]),
    "rpcc_contour_decode_wide": (C.c_int, [_VP, _VP, _I, _I, _I, _VP, _VP, _VP]),
    "rpcc_decode_wide": (C.c_int, [_VP, _VP, _VP, _VP, C.POINTER(C.c_double), _I, _VP, _I, _I, _I, _VP, _VP, _VP, _VP]),
    "rpcc_debug_stamps": (C.c_int, [_VP]),
    "rpcc_timer_create": (_VP, []),
    "rpcc_timer_destroy": (None, [_VP]),
    "rpcc_timer_reserve": (C.c_int, [_VP, C.c_int]),
    "rpcc_timer_read": (C.c_int, [_VP, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
}


HOST_LIB_PATH = os.path.join(_HERE, "lib", "librpcc_host.so")
_host = None


HOST_ABI = 101      # rpcc_host_version() this binding was written against (a stale .so after an interface change is refused)
_host_failed = None


def host_lib():
    """librpcc_host.so (include/rpcc_host.h): the host-side container packer.  Raises RpccError when it is not built, cannot
    be loaded (e.g. libbz2.so.1.0 missing at run time) or is a stale build; the failure is remembered, so the callers'
    fallback (compress_utils.pack_frames -> the interpreter's bz2 module, same bytes) is taken at once afterwards."""
    global _host, _host_failed
    if _host_failed is not None:
        raise RpccError(_host_failed)
    if _host is None:
        try:
            if not os.path.exists(HOST_LIB_PATH):
                raise OSError("not built; run `python -c 'import __graft_entry__ as g; g.build()'`")
            h = C.CDLL(HOST_LIB_PATH)
            h.rpcc_host_version.restype = C.c_int
            h.rpcc_host_version.argtypes = []
            if h.rpcc_host_version() != HOST_ABI:
                raise OSError("reports interface version %d, this binding needs %d (stale build)" % (h.rpcc_host_version(), HOST_ABI))
            h.rpcc_host_pack_bz2.restype = C.c_int
            h.rpcc_host_pack_bz2.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        except (OSError, AttributeError) as e:
            _host_failed = "librpcc_host.so (%s): %s" % (HOST_LIB_PATH, e)
            raise RpccError(_host_failed)
        _host = h
    return _host


def exported_symbols():
    return sorted(_SIGS)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RpccError("librpcc_hip.so is not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'`"
                            % LIB_PATH)
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        if h.rpcc_version() != ABI_VERSION:   # the structs carry no size field: a library of another layout must not be called
            raise RpccError("librpcc_hip.so (%s) reports interface version %d, this binding needs %d (stale build: rebuild with "
                            "`python -c 'import __graft_entry__ as g; g.build()'`)" % (LIB_PATH, h.rpcc_version(), ABI_VERSION))
        _lib = h
    return _lib


def check(rc):
    if rc != 0:
        raise RpccError("librpcc_hip: %s (code %d)" % (lib().rpcc_last_error().decode(), rc))


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL); the tensor must be contiguous and on a GPU."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RpccError("expected a GPU tensor; the HIP path has no CPU fallback")
    if not t.is_contiguous():
        raise RpccError("expected a contiguous tensor")
    return C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)

"""librpcc_host.so: the native container packer produces the bytes of the per-frame Python path (bz2.compress per array +
[int32 length | bytes] records, utils/compress_utils.py:167-179,199-214).  CPU only."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def cu():
    import __graft_entry__ as ge
    ge.build()
    import rpcc_amd  # noqa: F401
    from rpcc_amd import compress_utils
    return compress_utils


def _frames(rng, n, uniform):
    out = []
    for i in range(n):
        nnz = int(rng.integers(0, 120000)) if i else 0      # frame 0: empty residual stream
        K = int(rng.integers(2, 102))
        od = {"residual_quantized": rng.normal(0, 3, nnz).astype(np.int16),
              "contour_map": rng.integers(0, 256, 16384, dtype=np.uint8),
              "idx_sequence": rng.integers(0, 102, int(rng.integers(0, 4000))).astype(np.uint16),
              "plane_param": rng.normal(0, 1, (K, 4)).astype(np.float32)}
        if not uniform:
            od["salience_level"] = rng.integers(0, 4, K).astype(np.uint8)
        out.append(od)
    return out


@pytest.mark.parametrize("uniform", [True, False])
def test_native_container_equals_python_path(cu, uniform):
    rng = np.random.default_rng(11 + uniform)
    bc = cu.BasicCompressor(method_name="bzip2")
    frames = _frames(rng, 6, uniform)
    want = [cu.pack_bitstream(bc.compress_dict(od), uniform=uniform) for od in frames]
    got = cu.pack_frames(bc, frames, uniform=uniform)
    assert got == want
    # views into a larger buffer (what the loader hands over) and a single frame
    big = np.concatenate([f["residual_quantized"] for f in frames])
    o = np.cumsum([0] + [len(f["residual_quantized"]) for f in frames])
    views = [dict(f, residual_quantized=big[o[i]:o[i + 1]]) for i, f in enumerate(frames)]
    assert cu.pack_frames(bc, views[2:3], uniform=uniform) == want[2:3]
    assert cu.pack_frames(bc, [], uniform=uniform) == []
    # round trip through the container reader
    back = cu.unpack_bitstream(got[3], uniform=uniform)
    assert np.array_equal(np.frombuffer(bc.decompress(back["residual_quantized"]), np.int16), frames[3]["residual_quantized"])


def test_container_reader_names_truncated_and_foreign_files(cu):
    """unpack_bitstream: a file cut short, a length that runs past the end (or is negative) and a container of the other framework are ValueErrors
    that name the payload -- not a struct.error, a silently short slice or a failure inside the entropy decoder."""
    rng = np.random.default_rng(3)
    bc = cu.BasicCompressor(method_name="bzip2")
    blob = cu.pack_frames(bc, _frames(rng, 2, True), uniform=True)[1]
    assert set(cu.unpack_bitstream(blob, uniform=True)) == {"contour_map", "idx_sequence", "plane_param", "residual_quantized"}
    for cut in (0, 3, 4, 10, len(blob) // 2, len(blob) - 1):
        with pytest.raises(ValueError, match="bitstream"):
            cu.unpack_bitstream(blob[:cut], uniform=True)
    bad = bytearray(blob)
    bad[0:4] = (-5).to_bytes(4, "little", signed=True)
    with pytest.raises(ValueError, match="claims -5 bytes"):
        cu.unpack_bitstream(bytes(bad), uniform=True)
    with pytest.raises(ValueError, match="bitstream"):          # four payloads read as five
        cu.unpack_bitstream(blob, uniform=False)


def test_other_back_ends_take_the_python_path(cu):
    rng = np.random.default_rng(3)
    bc = cu.BasicCompressor(method_name="gzip")
    frames = _frames(rng, 2, True)
    got = cu.pack_frames(bc, frames, uniform=True)
    for blob, od in zip(got, frames):
        back = cu.unpack_bitstream(blob, uniform=True)
        assert np.array_equal(np.frombuffer(bc.decompress(back["contour_map"]), np.uint8), od["contour_map"])


def test_host_header_symbols_exported(cu):
    from rpcc_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "rpcc_host.h")).read()
    declared = sorted(set(re.findall(r"\b(rpcc_host_[a-z0-9_]+)\s*\(", hdr)))
    assert declared == ["rpcc_host_pack_bz2", "rpcc_host_version"]
    lib = ctypes.CDLL(_lib.HOST_LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert _lib.host_lib().rpcc_host_version() == _lib.HOST_ABI


def test_unusable_host_library_falls_back_to_python(cu, monkeypatch):
    """A library that is present but cannot be used (libbz2 missing at run time, a stale build): pack_frames takes the per-frame
    Python path -- same bytes -- and the failure is remembered instead of being raised inside pool threads again and again."""
    from rpcc_amd import _lib
    rng = np.random.default_rng(5)
    bc = cu.BasicCompressor(method_name="bzip2")
    frames = _frames(rng, 3, True)
    want = [cu.pack_bitstream(bc.compress_dict(od), uniform=True) for od in frames]
    monkeypatch.setattr(_lib, "_host", None)
    monkeypatch.setattr(_lib, "_host_failed", None)
    monkeypatch.setattr(_lib, "HOST_ABI", 9999)            # "stale .so after an interface change"
    assert cu.pack_frames(bc, frames, uniform=True) == want
    assert _lib._host_failed and "stale" in _lib._host_failed
    assert cu.pack_frames(bc, frames, uniform=True) == want
    monkeypatch.setattr(_lib, "_host_failed", None)
    monkeypatch.setattr(_lib, "HOST_LIB_PATH", "/nonexistent/librpcc_host.so")
    assert cu.pack_frames(bc, frames, uniform=True) == want


def test_host_pack_return_codes(cu):
    """Distinct codes: RPCC_HOST_ERR_ARG for a bad argument, -(16 + frame) for the frame that does not fit its region."""
    from rpcc_amd import _lib
    h = _lib.host_lib()
    a = np.arange(1000, dtype=np.uint8)
    ptrs = np.array([a.ctypes.data, a.ctypes.data], np.uint64)
    lens = np.array([a.nbytes, a.nbytes], np.uint32)
    out = np.empty(8192, np.uint8)
    out_len = np.zeros(2, np.uint32)
    offs = np.array([0, 4096, 4100], np.uint64)             # frame 1 gets 4 bytes
    assert h.rpcc_host_pack_bz2(2, 1, ptrs.ctypes.data, lens.ctypes.data, out.ctypes.data, offs.ctypes.data, out_len.ctypes.data) == -17
    assert h.rpcc_host_pack_bz2(2, 0, ptrs.ctypes.data, lens.ctypes.data, out.ctypes.data, offs.ctypes.data, out_len.ctypes.data) == -1
    assert h.rpcc_host_pack_bz2(2, 1, ptrs.ctypes.data, lens.ctypes.data, out.ctypes.data, None, out_len.ctypes.data) == -1

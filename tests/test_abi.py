"""The C-ABI library loads and exports every symbol include/rpcc_hip.h declares (no GPU needed)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()
    import rpcc_amd  # noqa: F401
    from rpcc_amd import _lib
    return _lib


def test_header_symbols_exported(built):
    hdr = open(os.path.join(ROOT, "include", "rpcc_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(rpcc_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 12
    lib = ctypes.CDLL(built.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(built.exported_symbols()) == declared


def test_version_and_workspace(built):
    lib = built.lib()
    assert lib.rpcc_version() == built.ABI_VERSION
    assert lib.rpcc_workspace_bytes(256, 64 * 2048, 100, 0) > 0
    assert lib.rpcc_workspace_bytes(0, 64 * 2048, 100, 0) == 0


def test_argument_errors_do_not_crash(built):
    lib = built.lib()
    rc = lib.rpcc_fps_xyz(0, 10, 5, None, None, None, None)
    assert rc == -1 and b"bad argument" in lib.rpcc_last_error()


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under r-pcc_amd/ may import, link or load it."""
    bad = re.compile(r"(^|\n)\s*(from|import)\s+oracle\b|liborpcc|oracle/_ref|oracle\.oracle")
    for dp, _, fs in os.walk(os.path.join(ROOT, "r-pcc_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                assert not bad.search(open(os.path.join(dp, f)).read()), (dp, f)


def test_cluster_num_limits_are_named():
    """The reference takes any cluster_num (cfgs/compressor.yaml:22; uint16 labels in the stream).  The batch front-end does too (above 254 through the
    uint16-label entries, up to 65 533); the mirror classes' per-stage entries keep labels in a byte and must refuse a larger value with the limit
    spelled out -- in the front-end, before anything touches the device -- not with a bare argument error."""
    import re
    import pytest
    import rpcc_amd  # noqa: F401
    from rpcc_amd import _lib, ops
    hdr = open(os.path.join(ROOT, "include", "rpcc_hip.h")).read()
    assert int(re.search(r"#define RPCC_MAX_CLUSTERS (\d+)", hdr).group(1)) == _lib.MAX_CLUSTERS == 254
    assert int(re.search(r"#define RPCC_MAX_CLUSTERS_WIDE (\d+)", hdr).group(1)) == _lib.MAX_CLUSTERS_WIDE == 65533
    assert ops.check_cluster_num(254) == 254 and ops.check_cluster_num(1) == 1 and ops.check_cluster_num(300) == 300 and ops.check_cluster_num(65533) == 65533
    assert not ops.is_wide(254) and ops.is_wide(255)
    for bad in (65534, 0):
        with pytest.raises(_lib.RpccError, match=r"cluster_num = %d.*<= 65533" % bad):
            ops.check_cluster_num(bad)
    for bad in (255, 300):
        with pytest.raises(_lib.RpccError, match=r"cluster_num = %d.*stage-by-stage.*<= 254.*uint8.*BatchCompressor" % bad):
            ops.check_cluster_num(bad, wide=False)
    from rpcc_amd.pipeline import BatchCompressor
    with pytest.raises(_lib.RpccError, match="cluster_num = 70000"):
        BatchCompressor(None, cluster_num=70000)

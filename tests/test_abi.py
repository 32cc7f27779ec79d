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
    # the stage seams that exist on uint16 labels (segmentation, point model, prediction, uniform quantiser): up to RPCC_MAX_CLUSTERS_MID
    assert int(re.search(r"#define RPCC_MAX_CLUSTERS_MID (\d+)", hdr).group(1)) == _lib.MAX_CLUSTERS_MID == 1022
    assert ops.check_cluster_num(300, stage="mid") == 300 and ops.check_cluster_num(1022, stage="mid") == 1022
    with pytest.raises(_lib.RpccError, match=r"cluster_num = 1023.*<= 1022.*BatchCompressor"):
        ops.check_cluster_num(1023, stage="mid")
    from rpcc_amd.pipeline import BatchCompressor
    with pytest.raises(_lib.RpccError, match="cluster_num = 70000"):
        BatchCompressor(None, cluster_num=70000)


def test_nothing_throws_across_the_abi(built):
    """include/rpcc_hip.h promises `int` status codes and no C++ exception across the boundary.  The only throwing operations in the library are the
    growth of its two host-side tables (kernel attributes, timer events): every push_back sits behind a reserve inside a try block or inside one
    itself, and the timer entry points answer with a status where there is no device at all (here) instead of terminating the process."""
    src = open(os.path.join(ROOT, "r-pcc_amd", "csrc", "rpcc_hip.hip")).read().splitlines()
    for i, line in enumerate(src):
        if "push_back(" in line and not line.lstrip().startswith("//"):
            ctx = "\n".join(src[max(0, i - 14):i + 1])
            assert "try {" in ctx, "push_back outside a try block / reserve: line %d" % (i + 1)
    lib = built.lib()
    lib.rpcc_timer_create.restype = ctypes.c_void_p
    t = ctypes.c_void_p(lib.rpcc_timer_create())
    assert t.value
    assert lib.rpcc_timer_reserve(t, (1 << 20) + 1) == -1            # RPCC_ERR_ARG
    rc = lib.rpcc_timer_reserve(t, 4)                                # no device here: a HIP status, not an exception (0 on a GPU box)
    assert rc in (0, -2), rc
    lib.rpcc_timer_destroy.argtypes = [ctypes.c_void_p]
    lib.rpcc_timer_destroy(t)


def test_a_failed_collect_frees_its_ring_slot():
    """pipeline.BatchCompressor keeps SLOTS buffer sets per batch size; a collect() that raises (stream error, host OOM in a copy) or a submit()
    whose caller gives up (discard) must free the slot, or the ring is exhausted after SLOTS such events."""
    import rpcc_amd  # noqa: F401
    from rpcc_amd.pipeline import BatchCompressor

    class Buf:
        in_flight = True

    class BadStream:
        def synchronize(self):
            raise RuntimeError("stream error")

    bc = object.__new__(BatchCompressor)
    b = Buf()
    with pytest.raises(RuntimeError, match="stream error"):
        bc.collect(dict(buf=b, stream=BadStream()))
    assert b.in_flight is False
    b.in_flight = True
    with pytest.raises(RuntimeError, match="stream error"):
        bc.discard(dict(buf=b, stream=BadStream()))
    assert b.in_flight is False


def test_cpu_pinning_only_with_explicit_local_ranks(monkeypatch):
    """utils.pin_rank_cpus slices the host's CPUs by LOCAL_RANK: only when the launcher named LOCAL_RANK and LOCAL_WORLD_SIZE, and not when the
    process already runs on a subset of the machine (bound by its launcher)."""
    import rpcc_amd  # noqa: F401
    from rpcc_amd import utils
    for k in ("LOCAL_RANK", "LOCAL_WORLD_SIZE", "RPCC_NO_AFFINITY", "RPCC_FORCE_AFFINITY"):
        monkeypatch.delenv(k, raising=False)
    assert utils.local_rank_env() is None
    monkeypatch.setenv("LOCAL_RANK", "1")
    assert utils.local_rank_env() is None
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "4")
    assert utils.local_rank_env() == (1, 4)
    got = {}
    monkeypatch.setattr(os, "cpu_count", lambda: 16)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(16)), raising=False)
    monkeypatch.setattr(os, "sched_setaffinity", lambda pid, cpus: got.update(cpus=list(cpus)), raising=False)
    assert utils.pin_rank_cpus(1, 4) == [4, 5, 6, 7] and got["cpus"] == [4, 5, 6, 7]
    assert utils.pin_rank_cpus(4, 4) is None                       # a rank outside its local world
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(4, 8)), raising=False)
    got.clear()
    assert utils.pin_rank_cpus(1, 4) is None and not got           # already bound to 4 of 16 CPUs: left alone
    monkeypatch.setenv("RPCC_FORCE_AFFINITY", "1")
    assert utils.pin_rank_cpus(1, 4) == [5]

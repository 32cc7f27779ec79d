"""N>1 path on CPU: two gloo processes shard a datalist round-robin and gather variable-length
payloads to rank 0 in datalist order (the only exchange step of the multi-GPU design)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _payload(i):
    return bytes([(i * 7 + k) % 251 for k in range(10 + (i * 37) % 90)])


def _worker(rank, world, port, n_items, q):
    sys.path.insert(0, ROOT)
    import rpcc_amd  # noqa: F401
    from rpcc_amd.sharding import gather_payloads, shard_indices
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_indices(n_items, rank, world)
    out = gather_payloads([_payload(i) for i in mine], torch.device("cpu"))
    if rank == 0:
        q.put([o == _payload(i) for i, o in enumerate(out)])
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [7, 8, 1])
def test_two_rank_gloo_gather(n_items):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + n_items
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert len(res) == n_items and all(res)


def _round_worker(rank, world, port, n_items, round_items, outdir, q):
    """tools/compress_datalist.py --gather with the device part replaced by a stand-in: every rank "compresses" its shard in
    batches of 3, hands the blobs to RoundGather as they finish, rank 0 writes the files it receives."""
    sys.path.insert(0, ROOT)
    import rpcc_amd  # noqa: F401
    from rpcc_amd.sharding import RoundGather, shard_indices
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_indices(n_items, rank, world)
    rg = RoundGather(n_items, rank, world, torch.device("cpu"), round_items=round_items)
    got = []
    for s in range(0, len(mine), 3):
        got += rg.add([_payload(i) for i in mine[s:s + 3]])
    got += rg.finish()
    for i, blob in got:
        with open(os.path.join(outdir, "%04d.rpcc" % i), "wb") as f:
            f.write(blob)
    if rank == 0:
        q.put(([i for i, _ in got], rg.rounds))
    else:
        assert got == []
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items,round_items", [(11, 2), (8, 4), (5, 100), (1, 1), (12, 3)])
def test_two_rank_round_gather_writes_single_rank_files(tmp_path, n_items, round_items):
    """The product's collective (sharding.RoundGather, used by tools/compress_datalist.py --gather): two gloo ranks, the files
    rank 0 writes equal those of a single-rank run, the indices arrive in datalist order, and every rank ran the same number of
    rounds."""
    ctx = mp.get_context("spawn")
    outs = {}
    for world in (1, 2):
        d = tmp_path / ("w%d" % world)
        d.mkdir()
        q = ctx.Queue()
        port = 29700 + 10 * n_items + world
        procs = [ctx.Process(target=_round_worker, args=(r, world, port, n_items, round_items, str(d), q)) for r in range(world)]
        for p in procs:
            p.start()
        idx, rounds = q.get(timeout=120)
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
        assert idx == list(range(n_items))
        outs[world] = {f: open(os.path.join(d, f), "rb").read() for f in sorted(os.listdir(d))}
    assert len(outs[1]) == n_items and outs[1] == outs[2]
    assert all(outs[2]["%04d.rpcc" % i] == _payload(i) for i in range(n_items))


def _region_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import rpcc_amd  # noqa: F401
    from rpcc_amd.sharding import agree_steps, gather_rank_times
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    # the ranks measured different warm-up step times: 0.8 ms and 0.5 ms -> 200 ms need 250 / 400 steps; everyone runs 400
    steps = agree_steps(20, 0.8e-3 if rank == 0 else 0.5e-3, 0.2, dev)
    same = agree_steps(1000, 1e-3, 0.2, dev)          # the request is already long enough
    nowarm = agree_steps(20, 0.0, 0.2, dev)           # no warm-up timing: the request stands
    times = gather_rank_times(0.1 * (rank + 1), dev)
    q.put((rank, steps, same, nowarm, times))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_timed_region_helpers():
    """bench.py at N > 1: all ranks agree on the number of timed steps (a collective runs per step) -- the longest any rank's
    warm-up asks for to fill 200 ms, never fewer than requested -- and every rank's own time arrives in rank order."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_region_worker, args=(r, 2, 29677, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, steps, same, nowarm, times in res:
        assert steps == 400 and same == 1000 and nowarm == 20
        assert [round(t, 3) for t in times] == [0.1, 0.2]


def test_shard_indices_partition():
    sys.path.insert(0, ROOT)
    import rpcc_amd  # noqa: F401
    from rpcc_amd.sharding import owner_of, shard_indices
    for n in (0, 1, 5, 16, 17):
        for w in (1, 2, 3, 8):
            parts = [shard_indices(n, r, w) for r in range(w)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert all(owner_of(i, w) == r for r in range(w) for i in parts[r])


def _stream_of(rank, frame):
    import numpy as np
    rng = np.random.default_rng(1000 * rank + frame)
    return rng.integers(-3000, 3000, int(rng.integers(0, 40)), dtype=np.int64).astype(np.int16)


def _exchange_worker(rank, world, port, frames, q):
    """What bench.py does per step at N > 1, with the device-side packing emulated on the CPU."""
    sys.path.insert(0, ROOT)
    import numpy as np
    import rpcc_amd  # noqa: F401
    from rpcc_amd.sharding import PackedExchange
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    runs = [_stream_of(rank, f) for f in range(frames)]
    points = sum(len(r) for r in runs) + 5 * (rank + 1)            # point counts differ between ranks; nnz <= points
    cap = PackedExchange.agree_capacity(points, dev)
    ex = PackedExchange(frames, cap, dev)
    ok = []
    for step in range(3):                                           # three steps reuse the receive buffers
        packed = torch.zeros((cap,), dtype=torch.int16)
        cat = np.concatenate([np.roll(r, step) for r in runs]) if runs else np.zeros(0, np.int16)
        packed[: cat.shape[0]] = torch.from_numpy(cat)
        nnz = torch.tensor([len(r) for r in runs], dtype=torch.int32)
        ex.step(packed, nnz)
        if rank == 0:
            for r in range(world):
                for f in range(frames):
                    ok.append(np.array_equal(ex.frame_stream(r, f).numpy(), np.roll(_stream_of(r, f), step)))
    # lengths-only exchange (bench.py's default): the payload stays with its rank, rank 0 learns every frame's length
    ex2 = PackedExchange(frames, 0, dev, payloads=False)
    nnz = torch.tensor([len(r) for r in runs], dtype=torch.int32)
    ex2.step(None, nnz)
    assert ex2.pay_all is None and ex2.bytes_per_step() == frames * 4 * world
    for r in range(world):
        ok.append(ex2.nnz_all[r].tolist() == [len(_stream_of(r, f)) for f in range(frames)])
    if rank == 0:
        q.put(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_packed_exchange():
    """bench.py's exchange step (sharding.PackedExchange: capacity agreement, all_gather of the lengths, gather of the packed
    int16 streams as bytes) between two gloo ranks: rank 0 can cut every frame's stream of every rank out of what it received."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_exchange_worker, args=(r, 2, 29655, 6, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert len(res) == 3 * 2 * 6 + 2 and all(res)


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus N` spawns its own N ranks (one per GPU) before touching the GPU; when fewer GPUs are visible it
    must fail loudly instead of measuring fewer."""
    import subprocess
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "GPU" in r.stderr and not r.stdout.strip().startswith("{")
    env2 = dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env2, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def _bench_env(**kw):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RPCC_RDZV_FILE"):
        env.pop(k, None)
    env.update(kw)
    return env


def test_bench_spawn_dry_run_two_ranks():
    """The launch path of `python bench.py --gpus 2` without a GPU (RPCC_BENCH_DRYRUN: gloo ranks): the parent counts no GPU
    through HIP, starts two fresh rank processes, they meet through the parent's file store, and rank 0's single JSON line
    -- carrying ranks_joined -- is relayed as the last line."""
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=_bench_env(RPCC_BENCH_DRYRUN="gloo"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = r.stdout.strip().splitlines()
    rec = json.loads(out[-1])
    assert rec["n_gpus"] == 2 and rec["ranks_joined"] == 2 and len(rec["per_rank_s"]) == 2
    assert sum(1 for ln in out if ln.startswith("{")) == 1, "exactly ONE JSON line on stdout"


def test_bench_spawn_stops_the_group_when_a_rank_dies():
    """Rank 1 dies before the rendezvous: rank 0 would sit in init_process_group until the collective timeout.  The parent
    polls every child, kills the group and exits non-zero within seconds."""
    import subprocess
    import time
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"],
                       env=_bench_env(RPCC_BENCH_DRYRUN="gloo", RPCC_BENCH_DIE_RANK="1", RPCC_BENCH_PG_TIMEOUT_S="600"),
                       capture_output=True, text=True, timeout=300)
    took = time.monotonic() - t0
    assert r.returncode != 0 and "rank 1 exited with code 7" in r.stderr, (r.returncode, r.stderr[-2000:])
    assert took < 60, took                      # import torch dominates; the rendezvous timeout is 600 s
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_spawn_wall_budget():
    """A rank that never finishes: the parent's wall budget ends the run."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"],
                       env=_bench_env(RPCC_BENCH_DRYRUN="gloo", RPCC_BENCH_HANG_RANK="0", RPCC_BENCH_WALL_S="8"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "wall budget" in r.stderr, (r.returncode, r.stderr[-2000:])


def test_bench_spawn_dry_run_four_ranks():
    """Four gloo ranks through the same launch path (the round-end scaling run uses 2, 4 and 8)."""
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=_bench_env(RPCC_BENCH_DRYRUN="gloo"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec["n_gpus"] == 4 and rec["ranks_joined"] == 4 and len(rec["per_rank_s"]) == 4


def test_bench_spawn_relays_a_failed_verification():
    """Rank 0 prints its line with "verified": false and exits 3: the parent relays the line and keeps the exit code (it used to
    discard both and return 1)."""
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"],
                       env=_bench_env(RPCC_BENCH_DRYRUN="gloo", RPCC_BENCH_DRY_VERIFY_FAIL="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec["verified"] is False and rec["ranks_joined"] == 2


def test_bench_spawn_signal_to_the_parent_stops_every_rank():
    """The ranks live in sessions of their own, so a SIGTERM to the parent (the driver's `timeout`, Ctrl-C) reaches none of them
    by itself: the parent's handler must stop them.  A hung rank is left running; the parent is signalled; no child survives."""
    import signal
    import subprocess
    import time
    import psutil
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"],
                         env=_bench_env(RPCC_BENCH_DRYRUN="gloo", RPCC_BENCH_HANG_RANK="1", RPCC_BENCH_WALL_S="600"),
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    kids = []
    t0 = time.monotonic()
    while time.monotonic() - t0 < 120 and len(kids) < 2:
        time.sleep(0.2)
        try:
            kids = psutil.Process(p.pid).children(recursive=True)
        except psutil.NoSuchProcess:
            break
    assert len(kids) == 2, "the parent did not start its two ranks"
    time.sleep(1.0)
    p.send_signal(signal.SIGTERM)
    out, err = p.communicate(timeout=60)
    assert p.returncode == 128 + signal.SIGTERM and "all ranks stopped" in err, (p.returncode, err[-2000:])
    gone, alive = psutil.wait_procs(kids, timeout=20)
    assert not alive, "rank processes survived the parent: %s" % alive


def test_bench_parent_never_imports_torch_before_spawning():
    """The parent of `--gpus N` must stay a process that never touched HIP: the GPU count comes from sysfs / the
    *_VISIBLE_DEVICES masks (utils.visible_gpus), and spawn_ranks itself imports neither torch nor the HIP library."""
    import ast
    src = open(os.path.join(ROOT, "bench.py")).read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "spawn_ranks"][0]
    names = {a.name for n in ast.walk(fn) if isinstance(n, ast.Import) for a in n.names} | \
            {n.module for n in ast.walk(fn) if isinstance(n, ast.ImportFrom)}
    assert "torch" not in names and not any(str(m).startswith("torch") for m in names), names
    sys.path.insert(0, ROOT)
    import rpcc_amd  # noqa: F401
    from rpcc_amd import utils
    old = dict(os.environ)
    try:
        os.environ["HIP_VISIBLE_DEVICES"] = "0,1,2"
        n = utils.visible_gpus()
        assert n is not None and n <= 3
        os.environ["HIP_VISIBLE_DEVICES"] = ""
        assert utils.visible_gpus() == 0
    finally:
        os.environ.clear()
        os.environ.update(old)


def test_bench_finds_its_pmc_numbers():
    """bench.py fills roofline.traffic / valu_wave_insts from profiles/pmc_current.json: the committed file must describe
    the default configuration and name the FPS kernel the way bench.py looks it up (a rename once left the fields null)."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pm = json.load(open(os.path.join(root, "profiles", "pmc_current.json")))
    assert pm["config"] == {"batch": 256, "geom": "64x2048", "clusters": 100, "config": 1, "input": False}
    keys = [k for k in pm["kernels"] if k.startswith(("fps_regtab_planar_kernel", "fps_regtab_kernel<true", "fps_tiled_kernel<true"))]
    assert len(keys) == 1, list(pm["kernels"])
    assert pm["kernels"][keys[0]]["traffic_bytes_per_launch"] > 0 and pm["kernels"][keys[0]]["valu_wave_insts_per_launch"] > 0
    assert pm["step_traffic_bytes"] > 0 and pm["step_valu_wave_insts"] > 0     # roofline.frac (VALU issue) / step_traffic_frac
    # the VALU roofline is priced in SIMD cycles of the measured 2 / 4 / 8-cycle classes (profiles/r04_valu_peak.md)
    assert 2.0 * pm["step_valu_wave_insts"] <= pm["step_valu_simd_cycles"] <= 4.5 * pm["step_valu_wave_insts"]
    assert all(2.0 <= k["valu_mean_cycles_static"] <= 8.0 for k in pm["kernels"].values())
    src = open(os.path.join(root, "bench.py")).read()
    assert 'startswith(("fps_regtab_planar_kernel", "fps_regtab_kernel<true", "fps_tiled_kernel<true"))' in src


def test_bench_refuses_counters_of_another_build(tmp_path, monkeypatch):
    """roofline.frac / traffic come from committed profiler passes: they describe a run only when the kernels are the ones that were
    profiled.  profiles/pmc_current*.json carry a sha256 of csrc/* + include/rpcc_hip.h (tools_profiles.py); bench.pmc_numbers() recomputes
    it and answers "stale" -- the JSON line then says "pmc_stale": true and carries no counter-derived number -- for any other tree."""
    import argparse
    import json
    import shutil
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import rpcc_amd  # noqa: F401
    from rpcc_amd.build import DEPS, source_digest
    import bench
    digest = source_digest()
    assert len(digest) == 64 and digest == source_digest()
    a = argparse.Namespace(config=1, input=None, fps_bruteforce=False, scene="default")
    prof = tmp_path / "profiles"
    prof.mkdir()
    pm = json.load(open(os.path.join(root, "profiles", "pmc_current.json")))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    # (1) the hash of this tree: the numbers are used
    pm["source_sha256"] = digest
    json.dump(pm, open(prof / "pmc_current.json", "w"))
    got = bench.pmc_numbers(a, 256, "64x2048", 100)
    assert got and not got.get("stale") and got["step_traffic"] == pm["step_traffic_bytes"] and got["valu"] > 0
    # (2) another hash, and (3) a file from before the hashes: stale, nothing but the reason
    for bad in ("0" * 64, None):
        if bad is None:
            pm.pop("source_sha256")
        else:
            pm["source_sha256"] = bad
        json.dump(pm, open(prof / "pmc_current.json", "w"))
        got = bench.pmc_numbers(a, 256, "64x2048", 100)
        assert got["stale"] is True and "re-run the PMC passes" in got["why"] and "traffic" not in got and "step_valu" not in got
    # (4) the digest follows the sources: one more byte in a kernel file changes it
    k = [d for d in DEPS if d.endswith("codec_kernels.h")][0]
    keep = k + ".keep"
    shutil.copy2(k, keep)      # (keeps the modification time: no rebuild of the library after the test)
    try:
        open(k, "a").write("\n")
        assert source_digest() != digest
    finally:
        shutil.move(keep, k)
    assert source_digest() == digest
    # another configuration than the profiled one: no numbers either way (as before)
    assert bench.pmc_numbers(a, 128, "64x2048", 100) is None
    src = open(os.path.join(root, "bench.py")).read()
    assert '"pmc_stale": bool(pmc_stale)' in src and "cpu_single" in src


def _stub_blob(name):
    import zlib
    c = zlib.crc32(name.encode())
    return b"RPCC" + name.encode() * (1 + c % 5) + c.to_bytes(4, "little")


class _StubStreaming:
    """Stands in for loader.StreamingCompressor (the device part) in the datalist driver: same constructor arguments and run()."""

    def __init__(self, bc, batch, depth, workers, pool, points_per_frame, ingest):
        assert ingest == "rows"          # a datalist of .bin names: the driver hands the PATHS over
        self.B = batch

    def run(self, batches, sink, entropy=True):
        n = 0
        for k, (frames, fids) in enumerate(batches):
            assert len(frames) == len(fids) <= self.B
            sink(k, [_stub_blob(str(f)) for f in frames])
            n += len(frames)
        return n


def _datalist_rank(rank, world, port, datalist, outdir, logdir):
    """One rank of `tools/compress_datalist.py --gather` as torchrun would start it (environment variables), gloo instead of RCCL, the
    device part replaced by the stub above."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), RPCC_DIST_BACKEND="gloo")
    sys.stdout = open(os.path.join(logdir, "rank%d.log" % rank), "w")
    import rpcc_amd  # noqa: F401
    from rpcc_amd.tools import compress_datalist as cd
    from rpcc_amd.tools.compress import make_parser
    a = make_parser(datalist=True).parse_args(["--datalist", datalist, "--output_dir", outdir, "--lidar", "VelodyneVLP16", "--gather",
                                               "--gather-round", "16", "--batch", "8", "--workers", "2", "--output"])
    cd.compress(a, streaming_factory=_StubStreaming)
    print("affinity %s" % ",".join(str(c) for c in sorted(os.sched_getaffinity(0))))
    sys.stdout.flush()


def test_eight_rank_datalist_gather_dry_run(tmp_path):
    """SURVEY 8e / configs[3] without hardware: EIGHT gloo ranks run the datalist driver with --gather on a 1 000-entry datalist (the
    device part stubbed).  Rank 0 writes every file exactly once, with the right bytes, in datalist order; the other ranks write nothing;
    with enough CPUs every rank pins itself to its own slice of them."""
    world, n = 8, 1000
    names = ["/data/kitti/seq%02d/velodyne/%06d.bin" % (i % 11, i) for i in range(n)]
    datalist = tmp_path / "list.txt"
    datalist.write_text("\n".join(names) + "\n")
    outdir, logdir = tmp_path / "out", tmp_path / "log"
    outdir.mkdir(); logdir.mkdir()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_datalist_rank, args=(r, world, 29911, str(datalist), str(outdir), str(logdir))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    logs = [open(os.path.join(logdir, "rank%d.log" % r)).read() for r in range(world)]
    written = [ln.split(" -> ")[0] for ln in logs[0].splitlines() if " -> " in ln and ln.endswith("bytes")]
    assert written == names                                   # rank 0: every entry once, in datalist order
    for r in range(1, world):
        assert not any(" -> " in ln and ln.endswith("bytes") for ln in logs[r].splitlines()), r
    files = []
    for root, _, fs in os.walk(outdir):
        files += [os.path.join(root, f) for f in fs]
    assert len(files) == n
    for name in names:
        out = os.path.join(str(outdir), name[1:]).replace("bin", "rpcc")
        assert open(out, "rb").read() == _stub_blob(name), name
    import re
    frames = [int(re.search(r"rank %d/8: (\d+) frames" % r, logs[r]).group(1)) for r in range(world)]
    assert frames == [125] * 8 and sum("gathered to rank 0" in lg for lg in logs) == world
    aff = [set(int(c) for c in re.search(r"affinity ([\d,]+)", lg).group(1).split(",")) for lg in logs]
    if len(os.sched_getaffinity(0)) >= world:                  # every rank on its own CPUs
        assert all(aff[i].isdisjoint(aff[j]) for i in range(world) for j in range(i)), aff


def test_bench_preflight_and_n_rank_line_fields(monkeypatch):
    """bench.py --gpus 8 --preflight is the rehearsal of an 8-rank run on a one-GPU box (no ranks spawned); an N > 1 line carries the roofline object
    and says `"cpu_baseline": null` explicitly (the baseline is rank 0's at N = 1 only)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--preflight", "--batch", "64"])
    a = bench.parse()
    assert a.preflight and a.gpus == 8 and a.batch == 64
    src = open(os.path.join(root, "bench.py")).read()
    assert 'if a.preflight:\n        return preflight(a)' in src                     # before any rank is spawned
    assert 'if world > 1:\n            out["cpu_baseline"] = None' in src
    # the sharding pieces the rehearsal leans on: every rank of 8 arrives at the same number of gather rounds, the shards partition the list
    from rpcc_amd.sharding import RoundGather, shard_indices
    rounds = {RoundGather(13386, r, 8, "cpu", round_items=4096).rounds for r in range(8)}
    assert rounds == {1}
    assert sorted(i for r in range(8) for i in shard_indices(13386, r, 8)) == list(range(13386))
    assert {RoundGather(100000, r, 8, "cpu", round_items=4096).rounds for r in range(8)} == {4}

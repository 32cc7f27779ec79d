#!/usr/bin/env python3
"""bench.py -- frames/s of the R-PCC per-frame compression hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (projection -> ground fit + mask -> FPS segmentation -> model ->
intra-prediction -> quantisation + ordered scatter; entropy coder and file I/O excluded) over one
batch of synthetic 64x2048 sweeps (BASELINE.json configs[1]: batch = 256 frames per GPU, uniform + FPS +
point model, accuracy 0.02).  Inputs are resident in HBM before the timed region.

N > 1: one process per GPU (torch.distributed / RCCL), frames sharded across ranks (weak scaling, no data-path
collective).  Started without a launcher (`python bench.py --gpus N`) the parent spawns the N rank processes itself --
before it touches the GPU -- and relays rank 0's JSON line; under `python -m torch.distributed.run` the ranks are the
launcher's.  Per step the ranks exchange what rank 0 needs: the per-frame payload lengths (all_gather);
`--gather-payloads` additionally gathers the packed residual streams to rank 0 (configs[3] read literally).
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# VALU: 1024 SIMDs x 2.4 GHz cycles per second.  A wave64 instruction occupies its SIMD for 2 cycles (fma / mul / add / sub f32, add / sub
# u32, and / or / xor, right shifts, mov), 4 cycles (compares, selects, min / max, conversions, 3-operand integer forms, packed fp32,
# fp64, DPP, lane operations) or 8 (fp32 transcendentals) -- measured per class by tools_dev/valu_peak.hip, profiles/r04_valu_peak.md.
# The step's VALU work is therefore counted in SIMD cycles: PMC SQ_INSTS_VALU per kernel x the mean cycles of the kernel's static
# instruction mix (profiles/pmc_current.json: step_valu_simd_cycles), and priced against the cycles the chip has.
VALU_SIMD_CYCLES_PER_S = 1024 * 2.4e9


def _flush_c_stdio():
    try:
        C.CDLL(None).fflush(None)
    except Exception:
        pass


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)   # three batches in flight: the drain at the end of the timed region costs
    ap.add_argument("--warmup", type=int, default=5)   # about one step, so very short runs under-report (5 steps: -8 %, 20: -2 %)
    ap.add_argument("--batch", type=int, default=256, help="frames per GPU per step")
    ap.add_argument("--geom", default=None, help="HxW (default 64x2048; 64x2000 with --input)")
    ap.add_argument("--lidar", default=None, help="geometry by registry name instead of --geom: a lidar type or a dataset of r-pcc_amd/dataset.py "
                    "(Velodyne64E, Velodyne32E, VelodyneVLP16, KITTI_test = 80x2000, ...); the synthetic sweeps follow its field of view")
    ap.add_argument("--clusters", type=int, default=100)
    ap.add_argument("--accuracy", type=float, default=0.02)
    ap.add_argument("--config", type=int, default=1, choices=(1, 2),
                    help="BASELINE.json configs[]: 1 = uniform + point model (headline), 2 = non-uniform + plane model")
    ap.add_argument("--input", default=None,
                    help="real sweep (.npz with 'xyz', .bin or .npy): the batch is this frame replicated with a per-copy yaw "
                         "rotation instead of the synthetic scene, points in the file's order (64x2000 unless --geom)")
    ap.add_argument("--input-shuffle", action="store_true", help="with --input: shuffle each copy's points (the synthetic frames are shuffled)")
    ap.add_argument("--cpu-sample", type=int, default=24, help="frames timed on the host for cpu_baseline (0 = skip)")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle check of the outputs after the timed region")
    ap.add_argument("--gather-payloads", action="store_true", help="N>1: also gather the packed residual streams to rank 0 every step")
    ap.add_argument("--no-gather", action="store_true", help="N>1: no exchange at all")
    ap.add_argument("--force-gather", action="store_true",
                    help="N=1: run the exchange step as well (single-rank RCCL group) -- exercises the N>1 code path on one GPU")
    ap.add_argument("--h2d", action="store_true",
                    help="copy the batch's points from pinned host memory to the device inside every step (PCIe-inclusive rate; "
                         "NOT the headline configuration, which has its inputs resident in HBM)")
    ap.add_argument("--fps-bruteforce", action="store_true",
                    help="run the brute-force FPS kernel (streams every candidate for every sample: the reference algorithm's "
                         "roofline case) instead of the exact tile-pruned one; same results")
    ap.add_argument("--scene", default="default", choices=("default", "shell", "noise", "corridor"),
                    help="synthetic scene: the headline's (default) or an adversarial input of the FPS pruning study (synth.make_frame)")
    ap.add_argument("--groundless", type=int, default=0, help="this many of the batch's sweeps (spread evenly) lose every return below z = -1.45 m: fewer than 800 "
                                                              "ground candidates, the fit runs on the whole cloud (segment_utils.py:105-106)")
    ap.add_argument("--preflight", action="store_true",
                    help="with --gpus N on a ONE-GPU box: no ranks are spawned; everything of an N-rank run that depends on the rank count is exercised with "
                         "the world faked at the sharding layer (frame ids and seeds of every rank, the agreed exchange capacity and its receive buffers, "
                         "agree_steps, the datalist gather's rounds) on a single-rank RCCL group, one verified step per virtual rank, and the per-rank host / "
                         "pinned-memory budget is printed")
    ap.add_argument("--pipeline", type=int, default=3, help="batches in flight (streams); 1 = strictly serial steps")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary measurements (configs[2] fused, the real sweep, the datalist feed) that follow the headline at N=1")
    ap.add_argument("--verify-frames", type=int, default=64, help="frames per pipeline slot checked against the oracle when no cpu_baseline leg runs")
    return ap.parse_args()


# ----------------------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: spawn the ranks and supervise them.  The parent never touches HIP: it counts the GPUs from
# sysfs / the *_VISIBLE_DEVICES masks (utils.visible_gpus) and never imports torch.  The children are fresh processes
# (no exec from a process that initialised the GPU).  Rendezvous goes through a file store in a private temporary
# directory (no "bind, close and hope the port stays free").  Every child is polled: the first one that exits non-zero
# -- or the wall budget running out -- ends the whole group at once instead of leaving the others in init_process_group /
# a collective until the RCCL timeout.
# ----------------------------------------------------------------------------------------------------------------------
class _Stopped(Exception):
    """Raised inside spawn_ranks by its SIGTERM / SIGINT / SIGHUP handler."""


def _die_with_parent():
    """preexec_fn of a rank process (Linux): SIGKILL this child when the parent that forked it dies, however it dies (a
    SIGKILL of the parent runs no handler and no finally block)."""
    try:
        C.CDLL(None, use_errno=True).prctl(1, 9, 0, 0, 0)   # PR_SET_PDEATHSIG, SIGKILL
    except Exception:
        pass


def spawn_ranks(n, argv=None, wall_s=None):
    import shutil
    import signal
    import tempfile
    import threading
    import rpcc_amd  # noqa: F401
    from rpcc_amd.utils import visible_gpus
    dry = os.environ.get("RPCC_BENCH_DRYRUN")          # CPU dry run of the launch path (gloo ranks, tests/test_sharding.py)
    have = visible_gpus()
    if not dry:
        if have is None and os.path.exists("/dev/kfd"):
            # a container with the device nodes passed through but no KFD topology in sysfs: the count is unknown, not zero --
            # the supervised ranks fail fast if a device is missing
            print("bench.py: cannot count the GPUs without touching HIP (no KFD topology in sysfs); starting %d ranks anyway" % n, file=sys.stderr)
        elif (have or 0) < n:                          # (no /dev/kfd at all: not a ROCm host, no GPU)
            print("bench.py: --gpus %d but only %d GPU(s) are visible" % (n, have or 0), file=sys.stderr)
            return 2
    wall_s = float(os.environ.get("RPCC_BENCH_WALL_S", "1500")) if wall_s is None else wall_s
    rdzv = tempfile.mkdtemp(prefix="rpcc_rdzv_")
    procs, out0 = [], []

    def stop_all():                                   # the ranks' own process groups: nothing of theirs survives
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except (ProcessLookupError, PermissionError):
                    p.kill()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                pass

    def on_signal(signum, frame):
        raise _Stopped(signum)
    # The ranks run in sessions of their own (a group kill of one rank must not take the parent along), so a signal sent to the
    # parent -- the driver's `timeout`, Ctrl-C, a closed terminal -- reaches none of them: the parent hands it on.
    old_handlers = {}
    if threading.current_thread() is threading.main_thread():
        for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
            old_handlers[sg] = signal.signal(sg, on_signal)
    reader = None
    rc_out = None
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), RPCC_RDZV_FILE=os.path.join(rdzv, "store"),
                       HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + (sys.argv[1:] if argv is None else argv), env=env,
                                          stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, start_new_session=True,
                                          preexec_fn=_die_with_parent))
        reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
        reader.start()
        t0, failed = time.monotonic(), None
        while failed is None:
            rcs = [p.poll() for p in procs]
            bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
            if bad == [(0, 3)]:
                # rank 0 printed its line with "verified": false and exited 3 (after the group was torn down): the other ranks are
                # finishing; the line is relayed and the exit code kept
                t1 = time.monotonic()
                while any(p.poll() is None for p in procs) and time.monotonic() - t1 < 30:
                    time.sleep(0.05)
                if all(p.poll() == 0 for p in procs[1:]):
                    rc_out = 3
                    break
                failed = "rank 0 reported a verification failure and another rank did not finish"
            elif bad:
                failed = "rank %d exited with code %d" % bad[0]
            elif all(rc == 0 for rc in rcs):
                break
            elif time.monotonic() - t0 > wall_s:
                failed = "wall budget of %.0f s exceeded" % wall_s
            else:
                time.sleep(0.05)
        if failed is not None:
            stop_all()
            reader.join(timeout=10)
            print("bench.py: %s; all ranks stopped (exit codes %s)" % (failed, [p.poll() for p in procs]), file=sys.stderr)
            if out0 and out0[0]:                    # what rank 0 had printed so far: diagnostics, on stderr (stdout carries only a valid line)
                print("bench.py: rank 0's output before the stop:\n" + out0[0].decode(errors="replace")[-4000:], file=sys.stderr)
            return 1
        reader.join(timeout=10)
    except _Stopped as e:
        stop_all()
        print("bench.py: signal %d; all ranks stopped" % e.args[0], file=sys.stderr)
        return 128 + int(e.args[0])
    except BaseException:
        stop_all()
        raise
    finally:
        for sg, h in old_handlers.items():
            signal.signal(sg, h)
        shutil.rmtree(rdzv, ignore_errors=True)
    all_lines = (out0[0] if out0 else b"").decode(errors="replace").splitlines()
    js = [i for i, ln in enumerate(all_lines) if ln.startswith("{")]
    if not js:
        print("bench.py: rank 0 printed no JSON line", file=sys.stderr)
        return 1
    rec = json.loads(all_lines[js[-1]])
    if rec.get("n_gpus") != n or rec.get("ranks_joined", n) != n:
        print("bench.py: asked for %d GPUs, %s ranks joined" % (n, rec.get("ranks_joined", rec.get("n_gpus"))), file=sys.stderr)
        return 1
    for i, ln in enumerate(all_lines):     # ONE JSON line, and it is the last line on stdout
        if i != js[-1]:
            print(ln)
    print(all_lines[js[-1]], flush=True)
    return rc_out or 0


def init_group(backend, rank, world, device_id=None):
    """Process group of the ranks: the parent's file store when bench.py spawned them, the launcher's env:// otherwise."""
    import datetime
    import torch.distributed as dist
    kw = dict(rank=rank, world_size=world, timeout=datetime.timedelta(seconds=float(os.environ.get("RPCC_BENCH_PG_TIMEOUT_S", "600"))))
    if device_id is not None:
        kw["device_id"] = device_id
    if os.environ.get("RPCC_RDZV_FILE"):
        dist.init_process_group(backend, init_method="file://" + os.environ["RPCC_RDZV_FILE"], **kw)
    else:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend, **kw)
    return dist


def dry_run_rank(a, rank, world):
    """RPCC_BENCH_DRYRUN=gloo: the launch path of an N-rank run without a GPU -- rendezvous, barrier, the max-over-ranks
    reduction and the per-rank gather, with a JSON line of the same outer shape.  RPCC_BENCH_DIE_RANK=r makes rank r die
    before the rendezvous (the supervision test)."""
    import torch
    if os.environ.get("RPCC_BENCH_DIE_RANK") == str(rank):
        os._exit(7)
    if os.environ.get("RPCC_BENCH_HANG_RANK") == str(rank):
        time.sleep(3600)
    dist = init_group("gloo", rank, world)
    dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    dt = time.perf_counter() - t0
    allt = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(allt, torch.tensor([dt], dtype=torch.float64))
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        bad = os.environ.get("RPCC_BENCH_DRY_VERIFY_FAIL") == "1"     # what a failed oracle check looks like from outside: the line, then exit 3
        print(json.dumps({"metric": "dry run (no GPU work)", "value": 0.0, "n_gpus": world, "ranks_joined": len(allt),
                          "per_rank_s": [round(float(t.item()), 4) for t in allt], "verified": not bad}), flush=True)
        if bad:
            sys.exit(3)


SETUP_ROUNDS = 8   # untimed runs of every pipeline slot before the warm-up steps (run_workload)
LEN_EVERY = 8      # N > 1, lengths-only exchange: steps whose per-frame payload lengths travel in one all_gather


def load_real_batch(path, ids, H, W, dev, shuffle=False):
    """The real sweep replicated: copy i = the frame rotated about z by a per-copy angle; the points keep the file's order (the
    scanner's: ring by ring) unless `shuffle`."""
    import numpy as np
    import torch
    if path.endswith(".npz"):
        xyz = np.load(path)["xyz"]
    elif path.endswith(".bin"):
        xyz = np.fromfile(path, dtype=np.float32).reshape(-1, 4)[:, :3]
    else:
        xyz = np.load(path)[:, :3]
    xyz = np.ascontiguousarray(xyz[:, :3], dtype=np.float32)
    frames = []
    for i in ids:
        rng = np.random.Generator(np.random.PCG64(77_000 + int(i)))
        a = 2 * np.pi * ((int(i) * 0.61803398875) % 1.0)
        c, s = np.float32(np.cos(a)), np.float32(np.sin(a))
        f = xyz[rng.permutation(xyz.shape[0])] if shuffle else xyz
        frames.append(np.stack([c * f[:, 0] - s * f[:, 1], s * f[:, 0] + c * f[:, 1], f[:, 2]], 1).astype(np.float32))
    offs = np.zeros(len(frames) + 1, np.int64)
    offs[1:] = np.cumsum([f.shape[0] for f in frames])
    return torch.from_numpy(np.concatenate(frames)).to(dev), torch.from_numpy(offs).to(dev)


def pmc_numbers(a, B, geom_s, M):
    """HBM-side bytes and wave-level VALU instructions per launch / per step from the committed PMC passes (rocprofv3 cannot
    run inside this process); only when those passes were taken on this very configuration."""
    try:
        name = "pmc_current.json" if a.config == 1 and not a.input else "pmc_current_c%d%s.json" % (a.config, "_real" if a.input else "")
        pm = json.load(open(os.path.join(ROOT, "profiles", name)))
        if pm["config"] != {"batch": B, "geom": geom_s, "clusters": M, "config": a.config, "input": bool(a.input)} or a.fps_bruteforce or a.scene != "default" or getattr(a, "groundless", 0):
            return None
        key = [k for k in pm["kernels"] if k.startswith(("fps_regtab_planar_kernel", "fps_regtab_kernel<true", "fps_tiled_kernel<true"))][0]   # template arguments vary
        # The counters were taken on one build of the kernels: they describe this run only if the sources are the same (sha256 over
        # csrc/* + include/rpcc_hip.h, written by tools_profiles.py).  Another tree -> "pmc_stale": true and no counter-derived number.
        from rpcc_amd.build import source_digest
        have, want = pm.get("source_sha256"), source_digest()
        if have != want:
            return dict(stale=True, kernel=key, why="profiles/%s was taken on sources %s, this tree is %s: re-run the PMC passes (tools_dev/round_profiles.sh, "
                                                    "tools_profiles.py)" % (name, (have or "without a recorded hash")[:12], want[:12]))
        return dict(kernel=key, traffic=pm["kernels"][key]["traffic_bytes_per_launch"],
                    valu=pm["kernels"][key].get("valu_wave_insts_per_launch"),
                    valu_cyc=pm["kernels"][key].get("valu_mean_cycles_dynamic", pm["kernels"][key].get("valu_mean_cycles_static")),
                    step_traffic=pm.get("step_traffic_bytes"), step_valu=pm.get("step_valu_wave_insts"), step_cycles=pm.get("step_valu_simd_cycles"),
                    step_cycles_static=pm.get("step_valu_simd_cycles_static_mix"), step_cycles_lo=pm.get("step_valu_simd_cycles_lo"),
                    step_cycles_hi=pm.get("step_valu_simd_cycles_hi"), cycles_source=pm.get("valu_cycles_source", "static mix"),
                    src="profiles/%s_pmc.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU and the SQ_INSTS_VALU_<class> counters, separate passes, "
                        "serial steps, x2 read correction) + profiles/isa_mix_current.json (cycles per class); constants of the committed profile, not "
                        "measured in this run" % pm.get("tag", "?"))
    except Exception:
        return None


def run_workload(a, ctx):
    """One measurement: W warm-up steps, K timed steps (barrier + synchronize on both sides, max over ranks), the oracle
    check of every pipeline slot after the timed region.  Returns the JSON record on rank 0 (None elsewhere)."""
    import numpy as np
    import torch
    from rpcc_amd import ops, synth
    rank, world, dev, dist = ctx["rank"], ctx["world"], ctx["dev"], ctx["dist"]

    geom_s = a.geom or ("64x2000" if a.input else "64x2048")
    H, W = (int(v) for v in geom_s.split("x"))
    hfov_deg, vmax_deg, vmin_deg = 360, 2.0, -24.9
    if getattr(a, "lidar", None):
        from rpcc_amd import dataset as reg
        from rpcc_amd.utils import load_yaml
        y = load_yaml(reg.__lidar_cfg__[a.lidar] if a.lidar in reg.__lidar_cfg__ else reg.__dataset_cfg__[a.lidar])
        H, W, hfov_deg, vmax_deg, vmin_deg = int(y["RANGE_IMAGE_HEIGHT"]), int(y["RANGE_IMAGE_WIDTH"]), y["HORIZONTAL_FOV"], y["VERTICAL_ANGLE_MAX"], y["VERTICAL_ANGLE_MIN"]
        geom_s = "%dx%d" % (H, W)
    hfov, vmax, vmin = hfov_deg * (np.pi / 180), vmax_deg * (np.pi / 180), vmin_deg * (np.pi / 180)
    geom = ops.make_geom(H, W, hfov, vmax, vmin)
    tm_np = ops.transform_map(H, W, hfov, vmax, vmin)
    P, M, B = H * W, a.clusters, a.batch
    acc = a.accuracy * 2
    general = a.config == 2            # non-uniform framework + plane model

    # this rank's batch: frame ids are disjoint across ranks (frame-sharded datalist)
    ids = list(range(rank * B, rank * B + B))
    if a.input:
        xyz, offs = load_real_batch(a.input, ids, H, W, dev, shuffle=a.input_shuffle)
    else:
        xyz, offs = synth.make_batch(ids, H, W, device=dev, scene=a.scene, vmax_deg=vmax_deg, vmin_deg=vmin_deg, hfov_deg=hfov_deg)
    if getattr(a, "groundless", 0):
        o_h = offs.cpu().numpy()
        sel = set(int(v) for v in np.linspace(0, B - 1, min(a.groundless, B)).round())
        fr = [xyz[o_h[i]:o_h[i + 1]] for i in range(B)]
        fr = [f[f[:, 2] > -1.45] if i in sel else f for i, f in enumerate(fr)]
        offs = torch.as_tensor(np.concatenate([[0], np.cumsum([f.shape[0] for f in fr])]).astype(np.int64), device=dev)
        xyz = torch.cat(fr, 0).contiguous()
    offs_host = offs.cpu().numpy()
    fid = torch.as_tensor(np.asarray(ids, np.int64), device=dev)
    tm = torch.from_numpy(tm_np).to(dev)
    # batches in flight (software pipeline, depth --pipeline): step n runs on stream n % depth with its own
    # output buffers, so the latency-bound kernels of one step (FPS, ground fit: one workgroup per frame)
    # overlap the throughput-bound kernels of the next.  Every step is complete when the timed region ends.
    depth = max(1, a.pipeline)
    bufs = [ops.BatchBuffers(B, geom, M, dev, general=general) for _ in range(depth)]
    gms_l = [torch.zeros((B, 4), dtype=torch.float64, device=dev) for _ in range(depth)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(depth)] if depth > 1 else [torch.cuda.current_stream()]
    timer = ops.FpsTimer()
    timer.reserve(max(a.steps, a.warmup) + 8)   # hipEventCreate stays out of the timed region
    nu = ops.nonuniform_cfg(acc) if general else None    # compressor.yaml defaults (levels 30/10/3/0, +0/.02/.04/.06 m)

    if a.h2d:
        xyz_host = xyz.cpu().pin_memory()
        xyz_l = [torch.empty_like(xyz) for _ in range(depth)]
    exchange = (world > 1 or a.force_gather) and not a.no_gather
    exch = exch_alt = None
    exchange_bytes = 0
    mode = [0]          # 0: the exchange the flags select (the headline), 1: the other one (timed after it, `exchange_modes`)
    if exchange:
        # the exchange step.  Default: what rank 0 needs to index the job -- the per-frame payload lengths
        # (the payload bytes stay with the rank that made them and writes its own .rpcc files, as
        # tools/compress_datalist.py does).  --gather-payloads: every rank also packs its frames' residual runs back to
        # back (rpcc_pack_payload: sum(nnz) <= its point count, so `cap` entries always fit) and rank 0 receives them.
        from rpcc_amd.sharding import PackedExchange
        cap = PackedExchange.agree_capacity(int(offs_host[-1] - offs_host[0]), dev)
        packed_l = [torch.zeros((max(cap, 1),), dtype=torch.int16, device=dev) for _ in range(depth)]
        # lengths only: the lengths of LEN_EVERY consecutive steps travel in ONE all_gather (fewer, larger collectives: a collective
        # per 0.75 ms step costs the chain more in cross-stream events than in bytes); with payloads: one gather per step
        exch_len = PackedExchange(LEN_EVERY * B, 0, dev, payloads=False)
        exch_pay = PackedExchange(B, cap, dev, payloads=True)
        exch, exch_alt = (exch_pay, exch_len) if a.gather_payloads else (exch_len, exch_pay)
        exchange_bytes = exch_pay.bytes_per_step() if a.gather_payloads else exch_len.bytes_per_step() // LEN_EVERY
        lens_ring = torch.zeros((2, LEN_EVERY, B), dtype=torch.int32, device=dev)

    step_no = [0]
    pack_tot = torch.zeros((1,), dtype=torch.int64, device=dev)
    exchange_note = None

    def run(k, src, timed=True):
        ops.compress_batch(src, offs, tm, gms_l[k], bufs[k], ground_threshold=0.1, acc=acc, ground_seed=0, frame_ids=fid,
                           fps_bruteforce=a.fps_bruteforce, timer=timer if timed else None,
                           model_method="plane" if general else "point", angle_threshold=75, plane_seed=0, nonuniform=nu)

    # The exchange of a step runs on a stream of its own behind an event of the step's stream, so the next batch of that pipeline
    # slot is not held up by the collective (RCCL kernels of a few microseconds each cost the chain ~6 % when they sit on it); a
    # slot's next step waits for the slot's previous exchange before it overwrites the lengths / the packed stream.
    exch_stream = torch.cuda.Stream(device=dev) if exchange else None
    step_done = [torch.cuda.Event() for _ in range(depth)] if exchange else None
    exch_done = [None] * depth      # payload mode: the slot's previous gather (its packed stream may be overwritten after it)
    ring_done = [None, None]        # lengths mode: the previous all_gather of each half of the lengths ring
    len_no = [0]                    # steps whose lengths sit in the ring since the last flush boundary

    def send_lengths(r):
        """all_gather of ring half r on the exchange stream, behind the last step of every pipeline slot"""
        with torch.cuda.stream(exch_stream):
            for ev in step_done:
                exch_stream.wait_event(ev)
            exch_len.step(None, lens_ring[r].view(-1))
            ring_done[r] = torch.cuda.Event()
            ring_done[r].record(exch_stream)

    def flush_exchange():
        """end of a timed region: the lengths of the steps since the last full ring half (inside the region)"""
        if exchange and not (exch_alt if mode[0] else exch).payloads and len_no[0] % LEN_EVERY:
            send_lengths((len_no[0] // LEN_EVERY) % 2)
            len_no[0] += LEN_EVERY - len_no[0] % LEN_EVERY

    def step():
        k = step_no[0] % depth
        step_no[0] += 1
        with torch.cuda.stream(streams[k]):
            src = xyz
            if a.h2d:
                xyz_l[k].copy_(xyz_host, non_blocking=True)
                src = xyz_l[k]
            ex = (exch_alt if mode[0] else exch) if exchange else None
            if ex is not None and ex.payloads and exch_done[k] is not None:
                streams[k].wait_event(exch_done[k])
            if ex is not None and not ex.payloads:     # the call writes its per-frame lengths straight into their place in the ring
                r, j = (len_no[0] // LEN_EVERY) % 2, len_no[0] % LEN_EVERY
                if j < depth and ring_done[r] is not None:
                    streams[k].wait_event(ring_done[r])     # (first touch of this half by this slot's stream since its last all_gather)
                bufs[k].nnz = lens_ring[r, j]
            run(k, src)
            if ex is not None and ex.payloads:
                ops.pack_payload(bufs[k].q16, bufs[k].nnz, packed=packed_l[k], capacity=cap, total=pack_tot)
                step_done[k].record(streams[k])
                with torch.cuda.stream(exch_stream):
                    exch_stream.wait_event(step_done[k])
                    ex.step(packed_l[k], bufs[k].nnz)
                    exch_done[k] = torch.cuda.Event()
                    exch_done[k].record(exch_stream)
            elif ex is not None:
                step_done[k].record(streams[k])
                len_no[0] += 1
                if j == LEN_EVERY - 1:
                    send_lengths(r)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up (untimed, like the allocations above): every pipeline slot is run SETUP_ROUNDS times so that stream creation,
    # first-use kernel attribute calls and page faults of its buffers are not billed to a timed step, and so that the GPU has
    # left its idle power state before the W warm-up steps: with W = 5 alone a 20-step region measures 3-4 % below the
    # steady state (0.825 against 0.795 ms per step; --warmup 30 shows the same).  Reported as config.setup_batches.
    for _ in range(SETUP_ROUNDS):
        for k in range(depth):
            with torch.cuda.stream(streams[k]):
                run(k, xyz, timed=False)
    torch.cuda.synchronize()
    if exchange:
        # one untimed trial of the exchange with a content check on rank 0 (its own frames must come back as they were
        # packed).  An error that every rank sees alike (an unsupported dtype, a missing backend feature) switches the
        # exchange off instead of killing the run -- and says so in the JSON line.
        try:
            for ex in (exch, exch_alt):
                if ex.payloads:
                    ops.pack_payload(bufs[0].q16, bufs[0].nnz, packed=packed_l[0], capacity=cap, total=pack_tot)
                    ex.step(packed_l[0], bufs[0].nnz)
                else:
                    lens_ring[0, 0].copy_(bufs[0].nnz)
                    ex.step(None, lens_ring[0].view(-1))
                torch.cuda.synchronize()
                if rank == 0:
                    assert torch.equal(ex.nnz_all[0][:B], bufs[0].nnz), "exchange returned other lengths than sent"
                    if ex.payloads:
                        for f in (0, B // 2, B - 1):
                            n = int(bufs[0].nnz[f])
                            assert torch.equal(ex.frame_stream(0, f), bufs[0].q16[f, :n]), "exchange returned other data than packed"
        except Exception as e:  # noqa: BLE001
            exchange = False
            exchange_note = "exchange disabled after its trial failed: %s" % (str(e).splitlines()[0][:200],)
    # Length of the timed region.  N = 1: exactly --steps.  N > 1: at least RPCC_BENCH_MIN_REGION_MS (200 ms) -- a 15 ms region
    # (20 steps) would let one late rank or one slow collective decide the "scaling efficiency"; the number of steps is agreed
    # on by all ranks (they run a collective per step) from the warm-up's own timing and REPORTED as `steps`.
    torch.cuda.synchronize()
    tw = time.perf_counter()
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    tw = (time.perf_counter() - tw) / max(a.warmup, 1)
    steps = a.steps
    if world > 1:
        from rpcc_amd.sharding import agree_steps
        steps = agree_steps(a.steps, tw if a.warmup > 0 else 0.0, float(os.environ.get("RPCC_BENCH_MIN_REGION_MS", "200")) * 1e-3, dev)
        timer.reserve(steps + 8)

    def timed_region(k_steps):
        """K steps between two barriers; returns (this rank's time, [every rank's time]).  A rank's time runs from the opening
        barrier to ITS OWN torch.cuda.synchronize() after its K-th step -- the closing barrier (synchronize + RCCL barrier +
        synchronize) is outside it; the job's time is the slowest rank's."""
        barrier()
        timer.read()
        t0 = time.perf_counter()
        for _ in range(k_steps):
            step()
        flush_exchange()
        torch.cuda.synchronize()
        dt_local = time.perf_counter() - t0
        barrier()
        rank_dt = [dt_local]
        if world > 1:
            from rpcc_amd.sharding import gather_rank_times
            rank_dt = gather_rank_times(dt_local, dev)
        return max(rank_dt), rank_dt

    dt, rank_dt = timed_region(steps)
    fps_ms, fps_n = timer.read()
    # N > 1: the OTHER exchange mode, timed right after the headline's with the same number of steps (SURVEY 8e names the gather
    # of the bitstreams; the default exchanges the lengths only): both are reported under `exchange_modes`
    exchange_modes = None
    if exchange and exch_alt is not None:
        head_mode = "lengths+payloads" if a.gather_payloads else "lengths"
        exchange_modes = {head_mode: {"value": round(world * B * steps / dt, 2), "ms_per_step": round(dt / steps * 1e3, 4),
                                      "exchange_bytes_per_step": exchange_bytes}}
        try:
            flush_exchange()
            mode[0] = 1
            for _ in range(min(5, steps)):
                step()
            dt2, _ = timed_region(steps)
            exchange_modes["lengths" if a.gather_payloads else "lengths+payloads"] = {
                "value": round(world * B * steps / dt2, 2), "ms_per_step": round(dt2 / steps * 1e3, 4), "exchange_bytes_per_step": exch_alt.bytes_per_step() // (1 if exch_alt.payloads else LEN_EVERY)}
        except Exception as e:  # noqa: BLE001
            exchange_modes["error"] = str(e).splitlines()[0][:200]
        finally:
            mode[0] = 0
        timer.read()

    buf = bufs[0]
    info = buf.info.cpu().numpy()
    n_left, nnz = info[:, 0].astype(np.int64), info[:, 2].astype(np.int64)
    n_in = np.diff(offs_host)
    # algorithmic bytes (stream-once model per stage)
    fps_bytes = 20.0 * (M - 1) * n_left.sum()
    b_alg = 12.0 * n_in.sum() + 92.0 * P * B + 2.0 * nnz.sum() + fps_bytes

    # ---- after the timed region: the outputs of EVERY pipeline slot against the CPU oracle (sampled frames) ------------
    from oracle import oracle as orc
    g_o = orc.LidarGeom(H=H, W=W, hfov_deg=hfov_deg, vmax_deg=vmax_deg, vmin_deg=vmin_deg)
    cfg_o = dict(orc.DEFAULT_CFG, accuracy=a.accuracy, cluster_num=M)
    from rpcc_amd.utils import available_cpus
    threads = available_cpus()   # affinity mask capped by the cgroup quota: the CPUs the host leg really gets
    want_cpu = a.cpu_sample > 0 and world == 1 and rank == 0
    S = min(B, max(a.cpu_sample, 16 * threads)) if want_cpu else min(B, a.verify_frames)   # ~10 s of CPU work for the baseline leg
    frames_h = [xyz[offs_host[i]:offs_host[i + 1]].cpu().numpy() for i in range(S)]
    oracle_out = [None] * S

    def oracle_frame(i):  # ground RANSAC (sequential form of the same specification) + the reference hot path
        ri = orc.project(frames_h[i], g_o)
        gm = orc.ground_model(ri, tm_np, seed=ids[i])
        if general:
            o = orc.compress_frame(frames_h[i], g_o, tm_np, gm, cfg_o, uniform=False, plane=dict(angle_deg=75, seed=0, frame=ids[i]))
        else:
            o = orc.compress_frame(frames_h[i], g_o, tm_np, gm, cfg_o)
        oracle_out[i] = dict(ri=o["range_image"], gm=gm, pix=o["fps_pix"], seg=o["seg_idx"].astype(np.uint16),   # (labels above 255: cluster_num > 254, the uint16 entries)
                             model=np.asarray(o["model_param"]).astype(np.float32), q=o["q"].astype(np.int16),
                             sal=o.get("salience"))
        return 1

    def oracle_stage_latency(k):
        """the C port stage by stage on ONE host thread, k of the same frames -> mean ms per frame and stage"""
        t = dict.fromkeys(("a2 projection", "a4 ground fit", "a3+a5 back-projection, residual, mask", "a6 FPS", "a7 assignment", "a8 model", "a10+a11 prediction, residual, quantiser"), 0.0)
        for i in range(k):
            c = [time.perf_counter()]
            tick = lambda: c.append(time.perf_counter())
            ri = orc.project(frames_h[i], g_o); tick()
            gm = orc.ground_model(ri, tm_np, seed=ids[i]); tick()
            pc = orc.backproject(ri, tm_np)
            mask = orc.vertical_residual(pc, gm) > cfg_o["ground_threshold"]
            pc_left = pc[np.where(mask)]; tick()
            centers = pc_left[orc.fps(pc_left, M)]; tick()
            seg = orc.assign(ri, pc, tm_np, gm, centers).astype(np.int64); tick()
            mp = orc.cluster_modeling_plane(pc, ri, seg, tm_np, 75, 0, ids[i]) if general else None
            mp = np.concatenate((np.asarray(gm, np.float64).reshape(1, 4), mp), 0) if general else orc.point_model_param(ri, seg, gm); tick()
            res = ri.reshape(H, W, 1) - orc.intra_predict(seg, mp, tm_np)
            orc.uniform_quantize(seg, res, acc); tick()
            for name, d in zip(t, np.diff(c)):
                t[name] += d
        return {"frames": k, "ms_per_frame": round(sum(t.values()) / k * 1e3, 2), "stages_ms": {n: round(v / k * 1e3, 3) for n, v in t.items()},
                "what": "C port of the reference's cpu=True path (oracle/), one thread, stage by stage" + (" (uniform quantiser timed in place of the non-uniform one)" if general else "")}

    verified, cpu_rate = None, None
    if not a.no_verify or want_cpu:
        from concurrent.futures import ThreadPoolExecutor
        orc.lib()
        oracle_frame(0)
        t1 = time.perf_counter()
        with ThreadPoolExecutor(threads) as ex:       # frame-parallel like the reference's --workers pool; ctypes releases the GIL
            list(ex.map(oracle_frame, range(S)))
        cpu_rate = S / (time.perf_counter() - t1)
    cpu_single = oracle_stage_latency(min(S, 4)) if want_cpu else None
    if not a.no_verify:
        verified = True
        why = None
        for k in range(depth):
            bk = bufs[k]
            ri_d, seg_d, pix_d = bk.ri[:S].cpu().numpy(), bk.seg[:S].cpu().numpy(), bk.cen_pix[:S].cpu().numpy()
            gm_d, q_d, nz_d, mo_d = gms_l[k][:S].cpu().numpy(), bk.q16[:S].cpu().numpy(), bk.nnz[:S].cpu().numpy(), bk.model[:S].cpu().numpy()
            for i in range(S):
                o = oracle_out[i]
                nrow = o["model"].shape[0]
                ok = (np.array_equal(ri_d[i].view(np.uint32), o["ri"].view(np.uint32)) and
                      np.array_equal(gm_d[i].view(np.uint64), np.asarray(o["gm"], np.float64).view(np.uint64)) and
                      np.array_equal(pix_d[i], o["pix"]) and np.array_equal(seg_d[i].reshape(-1).astype(np.uint16), o["seg"].reshape(-1)) and
                      np.array_equal(mo_d[i, :nrow].view(np.uint32), o["model"].view(np.uint32)) and
                      int(nz_d[i]) == o["q"].shape[0] and np.array_equal(q_d[i, :nz_d[i]], o["q"]))
                if ok and general and o["sal"] is not None:
                    ok = np.array_equal(bufs[k].salience[i, :nrow].cpu().numpy(), o["sal"].astype(np.uint8))
                if not ok:
                    verified, why = False, "slot %d frame %d differs from the oracle" % (k, i)
                    break
            if not verified:
                break
        if world > 1:
            t = torch.tensor([1 if verified else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            verified = bool(int(t.item()))
        if why and rank == 0:
            print("bench.py: VERIFICATION FAILED: " + why, file=sys.stderr)
    ctx["last"] = dict(oracle_out=oracle_out, frames_h=frames_h, ids=ids, S=S, geom=(H, W), tm_np=tm_np)

    out = None
    if rank == 0:
        frames_per_s = world * B * steps / dt
        fps_launch_ms = fps_ms / max(fps_n, 1)
        lt = fps_launch_ms * 1e-3
        stream_once = fps_bytes / lt / 1e9 if fps_n else 0.0
        pm = pmc_numbers(a, B, geom_s, M)
        fps_kernel = "fps_range_kernel (brute force)" if a.fps_bruteforce else (pm["kernel"] if pm else "fps_regtab_kernel")
        pmc_stale = pm.get("why") if pm and pm.get("stale") else None
        if pmc_stale:
            pm = None      # (every counter-derived field below stays null)
        step_s = dt / steps
        mean_cyc = (pm["step_cycles"] / pm["step_valu"]) if pm and pm.get("step_valu") and pm.get("step_cycles") else None
        valu_peak = VALU_SIMD_CYCLES_PER_S / mean_cyc if mean_cyc else None       # wave-instructions per second of THIS instruction mix
        step_valu_frac = pm["step_cycles"] / step_s / VALU_SIMD_CYCLES_PER_S if mean_cyc else None
        step_traffic_frac = pm["step_traffic"] / step_s / 1e9 / HBM_PEAK_GBS if pm and pm.get("step_traffic") else None
        workload = (("configs[%d]: batch=%d %s " + (a.lidar if getattr(a, "lidar", None) else "Velodyne-64E") + " frames (%dx%d) per GPU, %s, accuracy=%g, cluster_num=%d, ground plane by "
                     "seeded RANSAC inside the step") % (a.config, B, "real (%s, rotated copies, %s)" % (os.path.basename(a.input), "shuffled" if a.input_shuffle else "points in stored order") if a.input
                                                       else ("synthetic" if a.scene == "default" else "synthetic ADVERSARIAL scene '%s'" % a.scene),
                                                       H, W, "non-uniform + FPS + plane-model" if general else "uniform + FPS + point-model", a.accuracy, M))
        if getattr(a, "groundless", 0):
            workload += "; %d of the %d sweeps without any return below z = -1.45 m (ground fit on the whole cloud)" % (min(a.groundless, B), B)
        exch_s = ("no exchange" if not exchange else
                  ("RCCL all_gather of the per-frame payload lengths + gather of the packed pre-entropy residual streams to rank 0, every step" if a.gather_payloads
                   else "RCCL all_gather of the per-frame payload lengths, %d steps per collective (payload bytes stay with the rank that writes the files)" % LEN_EVERY))
        if world > 1 or a.force_gather:
            workload += "; exchange per step: " + exch_s
        # The roofline object.  The resource the step uses most is the VALU (profiles/r04_valu_peak.md: the throughput kernels keep the
        # pipes 48-79 % busy when alone, the pipelined step ~60 %; HBM-side traffic is at ~40 % of the spec), so `bound`, `achieved`,
        # `peak` and `frac` describe THAT resource over the whole step.  The task statement's stream-once HBM figure of the
        # dominant kernel (FPS) is kept under dominant_kernel.stream_once, labelled: the exact tile-pruned kernel does not
        # move those bytes, so it exceeds the peak and bounds nothing.
        roof = {"bound": "valu", "scope": "whole step (all launches of one batch, %d batches in flight)" % depth,
                "what": "VALU pipe time of the step / measured step time: wave-level VALU instructions per kernel (PMC SQ_INSTS_VALU) x the mean SIMD "
                        "cycles per instruction (2 / 4 / 8 cycles by class, measured: profiles/r04_valu_peak.md; the kernel's DYNAMIC class counts -- "
                        "SQ_INSTS_VALU_ADD_F32 ... -- weighted with the static cycles inside each class when the committed profile holds them, else the "
                        "static mix), summed over the step's launches, against 1024 SIMDs x 2.4 GHz; achieved / peak in wave-instructions of this mix",
                "pmc_stale": bool(pmc_stale), **({"pmc_stale_why": pmc_stale} if pmc_stale else {}),
                "cycles_from": (pm["cycles_source"] if pm else None),
                # the same instruction count priced differently: every instruction at the guide's 2 cycles / at 4 cycles, the static mix alone,
                # and every counter class at the least / largest cycles its static instructions have
                "frac_if": ({"all_2_cycles": round(2.0 * pm["step_valu"] / step_s / VALU_SIMD_CYCLES_PER_S, 4),
                             "all_4_cycles": round(4.0 * pm["step_valu"] / step_s / VALU_SIMD_CYCLES_PER_S, 4),
                             "static_mix": (round(pm["step_cycles_static"] / step_s / VALU_SIMD_CYCLES_PER_S, 4) if pm.get("step_cycles_static") else None),
                             "class_bounds": ([round(pm["step_cycles_lo"] / step_s / VALU_SIMD_CYCLES_PER_S, 4), round(pm["step_cycles_hi"] / step_s / VALU_SIMD_CYCLES_PER_S, 4)]
                                              if pm.get("step_cycles_lo") else None)} if mean_cyc else None),
                "achieved": (round(pm["step_valu"] / step_s / 1e9, 2) if mean_cyc else None),
                "peak": (round(valu_peak / 1e9, 1) if valu_peak else None), "unit": "G wave-instr/s",
                "frac": (round(step_valu_frac, 4) if step_valu_frac is not None else None),
                "mean_valu_cycles_per_instruction": (round(mean_cyc, 3) if mean_cyc else None),
                "traffic": (pm["step_traffic"] if pm else None),
                "step_valu_wave_insts": (pm["step_valu"] if pm else None),
                "step_valu_frac": (round(step_valu_frac, 4) if step_valu_frac is not None else None),
                "step_traffic_frac": (round(step_traffic_frac, 4) if step_traffic_frac is not None else None),
                "step_traffic_GBs": (round(pm["step_traffic"] / step_s / 1e9, 1) if pm and pm.get("step_traffic") else None),
                "pmc_source": pm["src"] if pm else None,
                "dominant_kernel": {
                    "kernel": fps_kernel, "launch_ms": round(fps_launch_ms, 4), "launches_timed": fps_n,
                    "timing": "HIP events on the launch stream around every FPS launch of the timed region",
                    "bound": "dependency chain: one workgroup per frame selects the M centres one after the other",
                    "critical_path": {"dependent_iterations": M - 1, "us_per_iteration": round(fps_launch_ms * 1e3 / max(M - 1, 1), 3)},
                    "traffic": pm["traffic"] if pm else None,
                    "traffic_frac": (round(pm["traffic"] / lt / 1e9 / HBM_PEAK_GBS, 4) if pm and fps_n else None),
                    "valu_wave_insts": pm["valu"] if pm else None,
                    "valu_frac": (round(pm["valu"] * pm["valu_cyc"] / lt / VALU_SIMD_CYCLES_PER_S, 4) if pm and pm.get("valu") and pm.get("valu_cyc") and fps_n else None),
                    "stream_once": {"model": "SURVEY 8d: 20*(M-1)*n_left bytes per frame, the bytes the reference's brute-force algorithm streams",
                                    "alg_bytes_per_launch": fps_bytes, "achieved": round(stream_once, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": round(stream_once / HBM_PEAK_GBS, 4),
                                    "note": "bounds nothing for the pruned kernel (frac > 1): the pruning is exact, results equal the "
                                            "brute-force kernel's bit for bit; --fps-bruteforce runs the kernel this model describes"}},
                "whole_path_stream_once_GBs": round(b_alg * steps / dt / 1e9, 2)}
        out = {
            "metric": "frames/s (64E, 64x2048 range img), projection->segmentation->model->quantise",
            "value": round(frames_per_s, 2), "unit": "frames/s", "n_gpus": world, "steps": steps, "warmup": a.warmup,
            "ms_per_step": round(dt / steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "real sweep replicated" if a.input else "synthetic",
            "verified": verified, "ranks_joined": len(rank_dt),
            "per_rank_frames_per_s": {"min": round(B * steps / max(rank_dt), 2), "max": round(B * steps / min(rank_dt), 2)},
            "config": {"workload": workload, "frames_per_gpu_per_step": B, "batches_in_flight": depth,
                       "inputs": ("copied from pinned host memory inside every step (PCIe-inclusive run, not the headline)" if a.h2d
                                  else "resident in HBM before the timed region"),
                       "sharding": "frames over ranks, no data-path collective; " + exch_s
                       + ("; " + exchange_note if exchange_note else ""),
                       "exchange": ("none" if not exchange else ("lengths+payloads" if a.gather_payloads else "lengths")),
                       "exchange_bytes_per_step": exchange_bytes if exchange else 0,
                       "verified_frames_per_slot": (S if verified is not None else 0),
                       "setup_batches": SETUP_ROUNDS * depth,    # untimed, before the W warm-up steps (see run_workload)
                       "timing": "per rank: opening barrier -> its own synchronize after the K-th step (the closing RCCL barrier is "
                                 "outside); job time = slowest rank"},
            "roofline": roof,
        }
        if steps != a.steps:
            out["steps_requested"] = a.steps
            out["config"]["min_timed_region_ms"] = float(os.environ.get("RPCC_BENCH_MIN_REGION_MS", "200"))
        if exchange_modes is not None:
            out["exchange_modes"] = exchange_modes
        if world > 1:
            out["cpu_baseline"] = None      # (rank 0 at N = 1 only: the task statement's rule; the roofline object above is carried at every N)
        if want_cpu:
            out["cpu_baseline"] = {"value": round(cpu_rate, 3), "unit": "frames/s", "cores": threads, "kind": "port",
                                   "sample": "%d of the same frames, C port of the reference cpu=True path "
                                             "(oracle/), frame-parallel over %d threads" % (S, threads),
                                   "single_thread": cpu_single}
    del bufs, gms_l, xyz
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return out


def run_feed(a, ctx, frames_total=8192, ingest="xyz"):
    """Secondary: the datalist feed (loader.StreamingCompressor) without the entropy coder -- frames in host memory ->
    pinned staging -> H2D -> device path + contour codec + payload packing -> D2H of the packed payload.  Verified: the
    residual streams of the first batch against the oracle outputs of the headline run (same frames, same seeds)."""
    import numpy as np
    import torch
    from rpcc_amd.loader import StreamingCompressor
    from rpcc_amd.pipeline import BatchCompressor
    from rpcc_amd.transformer import PCTransformer
    last = ctx["last"]
    H, W = last["geom"]
    B = a.batch
    T = PCTransformer(dict(HORIZONTAL_FOV=360, VERTICAL_ANGLE_MAX=2.0, VERTICAL_ANGLE_MIN=-24.9, RANGE_IMAGE_HEIGHT=H, RANGE_IMAGE_WIDTH=W),
                      device=str(ctx["dev"]))
    from rpcc_amd import synth
    base = [f.cpu().numpy() for f in (synth.make_frame(i, H, W, device=ctx["dev"]) for i in range(B))]
    if ingest == "rows":     # the sweeps as a .bin stores them: float32 rows (x, y, z, intensity), handed over unsliced
        base = [np.ascontiguousarray(np.concatenate([f, np.full((f.shape[0], 1), 0.5, np.float32)], 1)) for f in base]
    bc = BatchCompressor(T, cluster_num=a.clusters, accuracy=a.accuracy, seed=0)
    sc = StreamingCompressor(bc, batch=B, depth=4, ingest=ingest)
    nb = max(1, frames_total // B)

    def batches(n):
        for k in range(n):
            yield base, list(range(B))
    got = {}

    def sink(k, payload):
        if k == 0 and not got:
            got["q"] = [np.array(payload.frame(b)["residual_quantized"], copy=True) for b in range(min(last["S"], len(payload), 16))]
    sc.run(batches(2), sink=None, entropy=False)      # warm-up: pinned slots touched, streams created
    for k in sc.prof:
        sc.prof[k] = 0.0
    t0, c0 = time.perf_counter(), time.process_time()
    n = sc.run(batches(nb), sink=sink, entropy=False)
    torch.cuda.synchronize()
    dt, cpu = time.perf_counter() - t0, time.process_time() - c0
    ok = None
    if last["oracle_out"] and last["oracle_out"][0] is not None:
        ok = all(np.array_equal(q, last["oracle_out"][i]["q"]) for i, q in enumerate(got.get("q", []))) and len(got.get("q", [])) > 0
    return {"what": "datalist feed without the entropy coder: %d frames of %dx%d from host memory through loader.StreamingCompressor "
                    "(pinned staging, H2D, device path + contour codec + payload packing, D2H), batches of %d; ingest=%s: %s"
                    % (n, H, W, B, ingest, "first three columns staged by a host copy (strided for .bin rows), 12 B per point over the link" if ingest == "xyz" else
                       "the stored [N,4] rows copied whole, 16 B per point over the link, 16-byte row loads in the kernel"),
            "value": round(n / dt, 1), "unit": "frames/s", "verified": ok,
            "host_cpu_ms_per_frame": round(cpu / max(n, 1) * 1e3, 4), "host_stage_ms_per_batch": round(sc.prof["stage"] / max(nb, 1) * 1e3, 3),
            "bound": "host copy into pinned memory + PCIe H2D, not the kernels"}


def run_mixed(a, ctx, per=85, reps=24, slots=None, by_streams=False):
    """Secondary: configs[4]'s content on one GPU -- 64E / 32E / VLP16 sweeps (variable H x W) in mixed batches, non-uniform
    framework + plane model: every mixed batch is ONE fused call (rpcc_compress_batch_mixed: the per-frame and per-label kernels run
    once over the three geometry groups) plus the groups' contour coding, on the stream of its slot; `slots` mixed batches in
    flight, device part only.  by_streams (A/B, tools_dev): the form used until round 4 -- every group its own chain of launches on
    its own stream.  Verified: labels, salience levels and quantised integers of the first frames of every group against the oracle."""
    import numpy as np
    import torch
    from oracle import oracle as orc
    from rpcc_amd import dataset, synth
    from rpcc_amd.pipeline import BatchCompressor, MixedBatchCompressor
    dev = ctx["dev"]
    slots = int(os.environ.get("RPCC_MIXED_SLOTS", MixedBatchCompressor.SLOTS)) if slots is None else slots
    names = ("Velodyne64E", "Velodyne32E", "VelodyneVLP16")
    kw = dict(accuracy=a.accuracy, uniform=False, model_method="plane", seed=1)
    T = {n: dataset.build_dataset(lidar_type=n, device=str(dev)).PCTransformer for n in names}
    groups = []
    for n in names:
        gd = orc.GEOMS[n]
        ids = list(range(3000, 3000 + per))
        xyz, offs = synth.make_batch(ids, gd["H"], gd["W"], device=dev, vmax_deg=gd["vmax_deg"], vmin_deg=gd["vmin_deg"])
        sl = [(BatchCompressor(T[n], **kw), torch.cuda.Stream(device=dev)) for _ in range(slots)] if by_streams else None
        groups.append((n, gd, sl, xyz, offs, torch.as_tensor(np.asarray(ids, np.int64), device=dev)))
    mixed = [(MixedBatchCompressor(T, **kw), torch.cuda.Stream(device=dev)) for _ in range(slots)]
    parts = {n: (xyz, offs, None, fid) for n, gd, sl, xyz, offs, fid in groups}
    outs = {}

    def mixed_batch(r):
        if by_streams:           # its three geometry groups on the streams of slot r % slots
            for n, gd, sl, xyz, offs, fid in groups:
                bc, st = sl[r % slots]
                with torch.cuda.stream(st):
                    outs[n] = bc.compress_device(xyz, offs, frame_ids=fid)
        else:                    # one fused call on the stream of slot r % slots
            mc, st = mixed[r % slots]
            with torch.cuda.stream(st):
                outs.update(mc.compress_device(parts))
    for r in range(2 * slots):      # warm-up (per-stream allocator pools, first-use attributes)
        mixed_batch(r)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(reps):
        mixed_batch(r)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    ok = True
    nver = min(per, 32)
    for n, gd, sl, xyz, offs, fid in groups:
        buf, g_fit, bits, seq, nseq, sal = outs[n]
        g = orc.LidarGeom(**gd)
        tm = orc.transform_map(g)
        o_h = offs.cpu().numpy()
        for i in range(nver):
            f = xyz[o_h[i]:o_h[i + 1]].cpu().numpy()
            gm = orc.ground_model(orc.project(f, g), tm, seed=1 + 3000 + i)
            o = orc.compress_frame(f, g, tm, gm, dict(orc.DEFAULT_CFG, accuracy=a.accuracy), uniform=False,
                                   plane=dict(angle_deg=75, seed=1, frame=3000 + i))
            nz = int(buf.nnz[i])
            ok = ok and (np.array_equal(buf.seg[i].cpu().numpy(), o["seg_idx"].astype(np.uint8)) and nz == o["q"].shape[0] and
                         np.array_equal(buf.q16[i, :nz].cpu().numpy(), o["q"].astype(np.int16)) and
                         np.array_equal(sal[i, :o["salience"].shape[0]].cpu().numpy(), o["salience"].astype(np.uint8)))
    n_frames = 3 * per
    return {"what": "configs[4] on one GPU: mixed batches of %d + %d + %d sweeps of 64x2000 / 32x2250 / 16x1800, non-uniform + "
                    "plane-model, %s, %d mixed batch(es) in flight; device part"
                    % (per, per, per, "the three geometry groups of a batch on their own streams" if by_streams else
                       "one fused call per mixed batch (per-frame / per-label kernels once over the three geometry groups)", slots),
            "value": round(n_frames / dt, 1), "unit": "frames/s", "ms_per_mixed_batch": round(dt * 1e3, 3), "verified": bool(ok),
            "verified_frames_per_group": nver}


def run_secondary(a, ctx):
    """Driver-witnessed numbers beside the headline (outside its timed region, rank 0 at N=1): configs[2] as one fused
    call per batch on the KITTI-64E shape, the reference's example sweep replicated, and the datalist feed."""
    import copy
    sec = {}

    def brief(rec):
        return {k: rec[k] for k in ("value", "unit", "ms_per_step", "steps", "verified")} | {"workload": rec["config"]["workload"],
                                                                                             "verified_frames_per_slot": rec["config"]["verified_frames_per_slot"]}
    for name, fn in (("feed", run_feed), ("feed_rows", lambda a_, c_: run_feed(a_, c_, ingest="rows")), ("mixed_lidars", run_mixed)):
        try:
            sec[name] = fn(a, ctx)
        except Exception as e:  # noqa: BLE001
            sec[name] = {"error": str(e).splitlines()[0][:300] if str(e) else repr(e)}
    for name, kw in (("configs2_fused", dict(config=2, geom="64x2000", input=None)),
                     ("real_sweep", dict(config=1, geom=None, input=os.path.join(ROOT, "tests", "golden", "example_64E.npz"))),
                     # the reference's own KITTI_test table (dataset/__init__.py:21: 80 x 2000, 630 FPS tiles)
                     ("kitti_test_80x2000", dict(config=1, geom=None, input=None, lidar="KITTI_test")),
                     # the two inputs the exact kernels are slowest on: sweeps without ground returns (whole-cloud ground fit) and ranges that are independent from
                     # pixel to pixel (nothing for the FPS to prune)
                     ("groundless_8_of_256", dict(config=1, geom=None, input=None, groundless=8)),
                     ("scene_noise", dict(config=1, geom=None, input=None, scene="noise")),
                     # cluster_num is a free value of the reference's YAML (cfgs/compressor.yaml:22): 300 needs uint16 labels
                     ("clusters_300", dict(config=1, geom=None, input=None, clusters=300))):
        b = copy.copy(a)
        for k, v in kw.items():
            setattr(b, k, v)
        b.cpu_sample, b.steps, b.warmup, b.h2d, b.fps_bruteforce = 0, max(20, min(a.steps, 50)), 5, False, False
        b.groundless, b.scene, b.clusters = kw.get("groundless", 0), kw.get("scene", "default"), kw.get("clusters", a.clusters)
        if "lidar" not in kw:
            b.lidar = None
        try:
            if b.input and not os.path.exists(b.input):
                raise FileNotFoundError(b.input)
            sec[name] = brief(run_workload(b, ctx))
        except Exception as e:  # noqa: BLE001
            sec[name] = {"error": str(e).splitlines()[0][:300] if str(e) else repr(e)}
    return sec


def preflight(a):
    """First contact with an N-GPU node cannot be rehearsed on the one-GPU boxes this build has seen (SCALE has been skipped every round), so this runs
    every rank-count-dependent piece of the N-rank path HERE with the world faked: for each virtual rank r of N its frame ids (disjoint, seeds follow
    them), its batch on the device, one verified step; the exchange capacity all ranks would agree on and rank 0's receive buffers for N ranks
    (allocated for real); agree_steps on the virtual ranks' step times; the datalist gather's round count for every rank of N; a real single-rank RCCL
    group for the collectives' code path.  Prints ONE JSON line ("preflight": true; never a benchmark value)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    N, B = max(a.gpus, 1), a.batch
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29544")
    assert torch.cuda.is_available(), "bench.py --preflight needs one GPU"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    import rpcc_amd  # noqa: F401
    from rpcc_amd import ops, synth
    from rpcc_amd.sharding import PackedExchange, RoundGather, agree_steps, shard_indices
    from rpcc_amd.utils import available_cpus
    from oracle import oracle as orc
    H, W, M = 64, 2048, a.clusters
    hfov, vmax, vmin = 2 * np.pi, 2.0 * np.pi / 180, -24.9 * np.pi / 180
    geom = ops.make_geom(H, W, hfov, vmax, vmin)
    tm_np = ops.transform_map(H, W, hfov, vmax, vmin)
    tm = torch.from_numpy(tm_np).to(dev)
    g_o = orc.LidarGeom(H=H, W=W, hfov_deg=360, vmax_deg=2.0, vmin_deg=-24.9)
    buf = ops.BatchBuffers(B, geom, M, dev)
    gms = torch.zeros((B, 4), dtype=torch.float64, device=dev)
    seen, pts, step_s, ok = set(), [], [], True
    for r in range(N):                                   # what rank r of N would hold and do
        ids = list(range(r * B, r * B + B))
        assert not (seen & set(ids)), "frame ids of two ranks overlap"
        seen |= set(ids)
        xyz, offs = synth.make_batch(ids, H, W, device=dev)
        fid = torch.as_tensor(np.asarray(ids, np.int64), device=dev)
        pts.append(int(xyz.shape[0]))
        for _ in range(2):
            ops.compress_batch(xyz, offs, tm, gms, buf, ground_seed=0, frame_ids=fid, acc=a.accuracy * 2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            ops.compress_batch(xyz, offs, tm, gms, buf, ground_seed=0, frame_ids=fid, acc=a.accuracy * 2)
        torch.cuda.synchronize()
        step_s.append((time.perf_counter() - t0) / 5)
        o_h = offs.cpu().numpy()
        for i in (0, B - 1):                            # the rank's first and last frame against the oracle (seeds = frame ids)
            f = xyz[o_h[i]:o_h[i + 1]].cpu().numpy()
            gm = orc.ground_model(orc.project(f, g_o), tm_np, seed=ids[i])
            o = orc.compress_frame(f, g_o, tm_np, gm, dict(orc.DEFAULT_CFG, accuracy=a.accuracy, cluster_num=M))
            n = int(buf.nnz[i])
            ok = ok and n == o["q"].shape[0] and np.array_equal(buf.q16[i, :n].cpu().numpy(), o["q"].astype(np.int16)) and \
                np.array_equal(gms[i].cpu().numpy().view(np.uint64), np.asarray(gm, np.float64).view(np.uint64))
        del xyz, offs
    # the exchange: the capacity the N ranks would agree on (a MAX all_reduce: run on the real group with this process's largest value), rank 0's
    # receive buffers for N ranks allocated for real, one collective through RCCL
    cap = PackedExchange.agree_capacity(max(pts), dev)
    assert cap == max(pts)
    recv = [torch.empty((2 * cap,), dtype=torch.uint8, device=dev) for _ in range(N)]      # PackedExchange.pay_all of rank 0 in an N-rank job
    lens = [torch.empty((8 * B,), dtype=torch.int32, device=dev) for _ in range(N)]
    recv_bytes = sum(t.numel() * t.element_size() for t in recv + lens)
    ex = PackedExchange(B, cap, dev, payloads=True)
    packed = torch.zeros((cap,), dtype=torch.int16, device=dev)
    ops.pack_payload(buf.q16, buf.nnz, packed=packed, capacity=cap, total=torch.zeros((1,), dtype=torch.int64, device=dev))
    ex.step(packed, buf.nnz)
    torch.cuda.synchronize()
    ok = ok and torch.equal(ex.nnz_all[0], buf.nnz)
    del recv, lens
    steps = agree_steps(a.steps, max(step_s), 0.2, dev)
    # the datalist driver's gather (tools/compress_datalist.py --gather): every rank of N must arrive at the same number of rounds, and the shards
    # must partition the datalist (13 386 entries: the reference's largest list, data/*.txt)
    n_items = 13386
    rounds = {RoundGather(n_items, r, N, dev, round_items=4096).rounds for r in range(N)}
    shards = [shard_indices(n_items, r, N) for r in range(N)]
    assert len(rounds) == 1 and sorted(i for sh in shards for i in sh) == list(range(n_items))
    P = H * W
    per_rank_pinned = 4 * B * P * 16 + 4 * B * P * 3      # loader.StreamingCompressor: 4 slots of B frames, rows ingest (16 B per point) + payload staging
    cpus = available_cpus()
    out = {"preflight": True, "metric": "preflight of an N-rank run on one GPU (no benchmark value)", "value": None, "n_gpus_rehearsed": N, "frames_per_rank_per_step": B,
           "verified": bool(ok), "frame_ids": "rank r: [r*%d, (r+1)*%d): %d disjoint ids, seeds follow them" % (B, B, len(seen)),
           "points_per_rank": {"min": min(pts), "max": max(pts)}, "exchange_capacity_entries": cap,
           "rank0_receive_buffers_bytes": int(recv_bytes), "rank0_receive_buffers_note": "payload mode: %d ranks x 2 x capacity bytes + lengths; allocated here for real" % N,
           "single_gpu_step_ms_per_virtual_rank": [round(t * 1e3, 3) for t in step_s], "agree_steps": {"requested": a.steps, "agreed": steps, "min_region_ms": 200},
           "datalist_gather": {"entries": n_items, "rounds_every_rank": rounds.pop(), "round_items": 4096, "shard_sizes": [len(sh) for sh in shards]},
           "host_budget_per_rank": {"pinned_bytes": per_rank_pinned, "pinned_bytes_all_ranks": N * per_rank_pinned, "cpus_visible_here": cpus,
                                    "cpus_per_rank_if_sliced": cpus // N, "note": "tools/compress_datalist.py pins rank r to slice r of the CPUs when the launcher names LOCAL_RANK and LOCAL_WORLD_SIZE"},
           "device_memory": {"free_bytes_now": int(torch.cuda.mem_get_info(dev)[0]), "batch_workspace_bytes": int(buf.ws.numel())},
           "rccl": "single-rank group formed, all_gather + gather + all_reduce(MAX) executed"}
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps(out), flush=True)
    if not ok:
        sys.exit(3)


def main():
    a = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if a.preflight:
        return preflight(a)
    if env_world is None and a.gpus > 1:
        sys.exit(spawn_ranks(a.gpus))
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d" % (a.gpus, world), file=sys.stderr)
        sys.exit(2)
    if os.environ.get("RPCC_BENCH_DRYRUN"):
        return dry_run_rank(a, rank, world)

    import torch
    dist = None
    if world > 1 or a.force_gather:
        dist = init_group("nccl", rank, world, device_id=torch.device("cuda", local))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import rpcc_amd  # noqa: F401
    ctx = dict(rank=rank, world=world, dev=dev, dist=dist)
    out = run_workload(a, ctx)
    headline = a.config == 1 and not a.input and a.geom is None and a.scene == "default" and not (a.fps_bruteforce or a.h2d or a.force_gather)
    if out is not None and world == 1 and headline and not a.no_secondary:     # beside the headline only
        out["secondary"] = run_secondary(a, ctx)
        out["secondary_verified"] = not any(isinstance(v, dict) and v.get("verified") is False for v in out["secondary"].values())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        # the ONE JSON line must be the last thing on stdout: libraries write there through C stdio as well (RCCL prints its
        # version banner to stdout under NCCL_DEBUG=VERSION, and a pipe delays it until the buffer is flushed)
        _flush_c_stdio()
        print(json.dumps(out), flush=True)
    sec_bad = out is not None and [k for k, v in out.get("secondary", {}).items() if isinstance(v, dict) and v.get("verified") is False]
    if sec_bad:
        print("bench.py: VERIFICATION FAILED in secondary measurement(s): %s" % ", ".join(sec_bad), file=sys.stderr)
    if out is not None and (out.get("verified") is False or sec_bad):
        sys.exit(3)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- frames/s of the R-PCC per-frame compression hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (projection -> ground mask -> FPS segmentation -> point model ->
intra-prediction -> quantisation + ordered scatter; entropy coder and file I/O excluded) over one
batch of synthetic 64x2048 sweeps (BASELINE.json configs[1]: batch = 256 frames per GPU, uniform + FPS +
point model, accuracy 0.02).  Inputs are resident in HBM before the timed region.  N > 1: one process
per GPU (torch.distributed / RCCL), frames sharded across ranks (weak scaling, no data-path
collective); each step ends with the exchange of configs[3]: the frames' residual streams, packed back to
back on the device, are gathered to rank 0 over RCCL (inside the timed region, overlapped with the next steps).  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


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
    ap.add_argument("--geom", default="64x2048")
    ap.add_argument("--clusters", type=int, default=100)
    ap.add_argument("--accuracy", type=float, default=0.02)
    ap.add_argument("--cpu-sample", type=int, default=24, help="frames timed on the host for cpu_baseline (0 = skip)")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the RCCL payload gather")
    ap.add_argument("--force-gather", action="store_true",
                    help="N=1: run the exchange step as well (single-rank RCCL group) -- exercises the N>1 code path on one GPU")
    ap.add_argument("--h2d", action="store_true",
                    help="copy the batch's points from pinned host memory to the device inside every step (PCIe-inclusive rate; "
                         "NOT the headline configuration, which has its inputs resident in HBM)")
    ap.add_argument("--fps-bruteforce", action="store_true",
                    help="run the brute-force FPS kernel (streams every candidate for every sample: the reference algorithm's "
                         "roofline case) instead of the exact tile-pruned one; same results")
    ap.add_argument("--slices", type=int, default=None, help="sub-batches on internal streams (library default 1)")
    ap.add_argument("--pipeline", type=int, default=3, help="batches in flight (streams); 1 = strictly serial steps")
    return ap.parse_args()


def cpu_baseline(frames, gms, g, tm, cfg, threads):
    """The CPU oracle (plain-C port of the reference's cpu=True path, validated bit-exact against the
    reference) on a bounded sample, frame-parallel like the reference's --workers ThreadPool
    (tools/compress_datalist.py:202-206).  ctypes releases the GIL inside the C calls."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as orc
    orc.lib()
    def run(i):   # ground RANSAC (sequential form of the same specification) + the reference hot path
        gm = orc.ground_model(orc.project(frames[i], g), tm, seed=i)
        return orc.compress_frame(frames[i], g, tm, gm, cfg)["q"].shape[0]
    run(0)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(run, range(len(frames))))
    dt = time.perf_counter() - t0
    return len(frames) / dt


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d -- N > 1 needs one process per GPU (python -m torch.distributed.run "
              "--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N); measuring %d GPU(s)" % (a.gpus, world, world),
              file=sys.stderr)
    if world > 1 or a.force_gather:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import rpcc_amd  # noqa: F401
    from rpcc_amd import ops, synth, _lib

    H, W = (int(v) for v in a.geom.split("x"))
    hfov, vmax, vmin = 360 * (np.pi / 180), 2.0 * (np.pi / 180), -24.9 * (np.pi / 180)
    geom = ops.make_geom(H, W, hfov, vmax, vmin)
    tm_np = ops.transform_map(H, W, hfov, vmax, vmin)
    P, M, B = H * W, a.clusters, a.batch
    acc = a.accuracy * 2

    # synthetic batch for this rank: frame ids are disjoint across ranks (frame-sharded datalist)
    ids = range(rank * B, rank * B + B)
    xyz, offs = synth.make_batch(ids, H, W, device=dev)
    offs_host = offs.cpu().numpy()      # frame boundaries are host knowledge (file sizes); enables sub-batch streams
    if a.slices is not None:
        ops.set_batch_slices(a.slices)
    if a.fps_bruteforce:
        ops.fps_force_bruteforce(True)
    tm = torch.from_numpy(tm_np).to(dev)
    # Two batches in flight (software pipeline, depth --pipeline): step n runs on stream n % depth with its own
    # output buffers, so the latency-bound kernels of one step (FPS, ground fit: one workgroup per frame)
    # overlap the throughput-bound kernels of the next.  Every step is complete when the timed region ends.
    depth = max(1, a.pipeline)
    bufs = [ops.BatchBuffers(B, geom, M, dev) for _ in range(depth)]
    gms_l = [torch.zeros((B, 4), dtype=torch.float64, device=dev) for _ in range(depth)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(depth)] if depth > 1 else [torch.cuda.current_stream()]
    buf, gms = bufs[0], gms_l[0]

    if a.h2d:
        xyz_host = xyz.cpu().pin_memory()
        xyz_l = [torch.empty_like(xyz) for _ in range(depth)]
    gather = (world > 1 or a.force_gather) and not a.no_gather
    if gather:
        # the exchange step: per step every rank packs its frames' residual runs back to back
        # (rpcc_pack_payload: sum(nnz) <= its point count, so `cap` entries always fit) and rank 0 receives the packed
        # streams + the per-frame lengths.  cap = the largest rank's point count, agreed on once, outside the timing.
        import torch.distributed as dist
        from rpcc_amd.sharding import PackedExchange
        cap = PackedExchange.agree_capacity(int(offs_host[-1] - offs_host[0]), dev)
        packed_l = [torch.zeros((cap,), dtype=torch.int16, device=dev) for _ in range(depth)]
        exch = PackedExchange(B, cap, dev)

    step_no = [0]
    pack_tot = torch.zeros((1,), dtype=torch.int64, device=dev)
    exchange_note = None

    def step():
        k = step_no[0] % depth
        step_no[0] += 1
        with torch.cuda.stream(streams[k]):
            src = xyz
            if a.h2d:
                xyz_l[k].copy_(xyz_host, non_blocking=True)
                src = xyz_l[k]
            ops.compress_batch(src, offs, tm, gms_l[k], bufs[k], ground_threshold=0.1, acc=acc, ground_seed=rank * B,
                               offsets_host=offs_host)
            if gather:
                ops.pack_payload(bufs[k].q16, bufs[k].nnz, packed=packed_l[k], capacity=cap, total=pack_tot)
                exch.step(packed_l[k], bufs[k].nnz)

    def barrier():
        torch.cuda.synchronize()
        if world > 1 or a.force_gather:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up (untimed, like the allocations above): touch every pipeline slot once so that stream creation, first-use
    # kernel attribute calls and page faults of its buffers are not billed to a timed step
    for k in range(depth):
        with torch.cuda.stream(streams[k]):
            ops.compress_batch(xyz, offs, tm, gms_l[k], bufs[k], ground_threshold=0.1, acc=acc, ground_seed=rank * B,
                               offsets_host=offs_host)
    torch.cuda.synchronize()
    if gather:
        # one untimed trial of the exchange with a content check on rank 0 (its own frames must come back as they were
        # packed).  An error that every rank sees alike (an unsupported dtype, a missing backend feature) switches the
        # exchange off instead of killing the run -- and says so in the JSON line.
        try:
            ops.pack_payload(bufs[0].q16, bufs[0].nnz, packed=packed_l[0], capacity=cap, total=pack_tot)
            exch.step(packed_l[0], bufs[0].nnz)
            torch.cuda.synchronize()
            if rank == 0:
                for f in (0, B // 2, B - 1):
                    n = int(bufs[0].nnz[f])
                    assert torch.equal(exch.frame_stream(0, f), bufs[0].q16[f, :n]), "exchange returned other data than packed"
        except Exception as e:  # noqa: BLE001
            gather = False
            exchange_note = "exchange disabled after its trial failed: %s" % (str(e).splitlines()[0][:200],)
    for _ in range(a.warmup):
        step()
    barrier()
    lib = _lib.lib()
    lib.rpcc_fps_timing(1)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    fps_ms, fps_n = C.c_double(0), C.c_int(0)
    lib.rpcc_fps_time_ms(C.byref(fps_ms), C.byref(fps_n))
    lib.rpcc_fps_timing(0)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    info = buf.info.cpu().numpy()
    n_left, nnz = info[:, 0].astype(np.int64), info[:, 2].astype(np.int64)
    n_in = np.diff(offs.cpu().numpy())
    # algorithmic bytes (stream-once model per stage)
    fps_bytes = 20.0 * (M - 1) * n_left.sum()
    b_alg = 12.0 * n_in.sum() + 92.0 * P * B + 2.0 * nnz.sum() + fps_bytes

    out = None
    if rank == 0:
        frames_per_s = world * B * a.steps / dt
        # the FPS kernel is launched once per sub-batch: average launch = (algorithmic bytes of its frames) / its time
        fps_launch_ms = fps_ms.value / max(fps_n.value, 1)
        launches_per_step = max(fps_n.value, 1) / a.steps
        fps_bytes_launch = fps_bytes / launches_per_step
        achieved = fps_bytes_launch / (fps_launch_ms * 1e-3) / 1e9 if fps_n.value else 0.0
        # HBM-side bytes per FPS launch from the committed PMC pass (rocprofv3 cannot run inside this process);
        # only reported when that pass was taken on this very configuration
        traffic, traffic_src = None, None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc.json")))
            if pm["config"] == {"batch": B, "geom": a.geom, "clusters": M} and not a.fps_bruteforce:
                key = [k for k in pm["kernels"] if k.startswith("fps_tiled_kernel<true")][0]   # template arguments vary
                traffic = pm["kernels"][key]["traffic_bytes_per_launch"] / pm.get("launches_per_step", 1)
                traffic_src = "profiles/%s_pmc.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, x2 read correction)" % pm.get("tag", "r01_v14")
        except Exception:
            pass
        out = {
            "metric": "frames/s (64E, 64x2048 range img), projection->segmentation->model->quantise",
            "value": round(frames_per_s, 2), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: batch=%d synthetic Velodyne-64E frames (%dx%d) per GPU, uniform + FPS + "
                                   "point-model, accuracy=%g, cluster_num=%d, ground plane by seeded RANSAC inside the step" % (B, H, W, a.accuracy, M),
                       "frames_per_gpu_per_step": B, "batches_in_flight": depth,
                       "inputs": ("copied from pinned host memory inside every step (PCIe-inclusive run, not the headline)" if a.h2d
                                  else "resident in HBM before the timed region"),
                       "sharding": "frames over ranks, no data-path collective"
                       + (", per step RCCL all_gather of the frame lengths + gather of the packed residual streams to rank 0" if gather else "")
                       + ("; " + exchange_note if exchange_note else "")},
            "roofline": {"bound": "hbm", "kernel": "fps_range_kernel (brute force)" if a.fps_bruteforce else "fps_tiled_kernel", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": traffic_src,
                         "traffic_GBs": (round(traffic / (fps_launch_ms * 1e-3) / 1e9, 2) if traffic else None),
                         "launch_ms": round(fps_launch_ms, 4), "alg_bytes_per_launch": fps_bytes_launch,
                         "launches_per_step": launches_per_step,
                         "whole_path_alg_GBs": round(b_alg * a.steps / dt / 1e9, 2),
                         "note": "achieved = algorithmic bytes of the brute-force stream model (20*(M-1)*n_left per frame, "
                                 "SURVEY 8d) / measured launch time; the exact tile-pruned kernel skips most of those bytes, "
                                 "so frac > 1 is expected; traffic = PMC-measured HBM-side bytes per launch"},
        }
        if a.cpu_sample > 0 and world == 1:
            S = min(B, max(a.cpu_sample, 2 * (os.cpu_count() or 1)))
            o = offs.cpu().numpy()
            frames = [xyz[o[i]:o[i + 1]].cpu().numpy() for i in range(S)]
            from oracle import oracle as orc
            g = orc.LidarGeom(H=H, W=W, hfov_deg=360, vmax_deg=2.0, vmin_deg=-24.9)
            cfg = dict(orc.DEFAULT_CFG, accuracy=a.accuracy, cluster_num=M)
            threads = os.cpu_count() or 1
            v = cpu_baseline(frames, gms.cpu().numpy(), g, tm_np, cfg, threads)
            out["cpu_baseline"] = {"value": round(v, 3), "unit": "frames/s", "cores": threads, "kind": "port",
                                   "sample": "%d of the same synthetic frames, C port of the reference cpu=True path "
                                             "(oracle/), frame-parallel over %d threads" % (S, threads)}
    if world > 1 or a.force_gather:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        # the ONE JSON line must be the last thing on stdout: libraries write there through C stdio as well (RCCL prints its
        # version banner to stdout under NCCL_DEBUG=VERSION, and a pipe delays it until the buffer is flushed)
        _flush_c_stdio()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

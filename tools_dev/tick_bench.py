"""GPU box: a DETERMINISTIC schedule of the batches' stages against bench.py's three independent streams.
A batch's chain alternates pixel-parallel and one-workgroup-per-frame stages:  T1 pix + band -> L1 ground fit -> T2 mask -> L2 FPS -> T3 assign, histogram,
scan, quantiser.  Tick t issues  L-stream: ground fit(b[t-1]), FPS(b[t-3]);  Ta-stream: pix + band(b[t]), mask(b[t-2]);  Tb-stream: T3(b[t-4])  -- every
launch of a tick depends on the tick before only, so the three streams meet at tick boundaries (events) and a latency kernel never runs beside two other
latency kernels.  One batch completes per tick; five batches in flight.   usage: python tools_dev/tick_bench.py [ticks]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpcc_amd  # noqa: E402,F401
from rpcc_amd import ops, synth  # noqa: E402

dev = torch.device("cuda:0")
H, W, M, B = 64, 2048, 100, 256
hfov, vmax, vmin = 2 * np.pi, 2.0 * np.pi / 180, -24.9 * np.pi / 180
geom = ops.make_geom(H, W, hfov, vmax, vmin)
tm = torch.from_numpy(ops.transform_map(H, W, hfov, vmax, vmin)).to(dev)
ids = list(range(B))
xyz, offs = synth.make_batch(ids, H, W, device=dev)
fid = torch.as_tensor(np.asarray(ids, np.int64), device=dev)
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 200
D = 5
bufs = [ops.BatchBuffers(B, geom, M, dev) for _ in range(D)]
gms = [torch.zeros((B, 4), dtype=torch.float64, device=dev) for _ in range(D)]
S = ops
kw = dict(ground_seed=0, frame_ids=fid)


def stages(mask, k):
    ops.compress_batch_stages(mask, xyz, offs, tm, gms[k], bufs[k], **kw)


def run(variant, n):
    sl, sa, sb, sl2 = (torch.cuda.Stream(device=dev) for _ in range(4))
    all_s = (sl, sa, sb, sl2)
    ev = {s: torch.cuda.Event() for s in all_s}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(n):
        prev = {s: torch.cuda.Event() for s in all_s}
        # tick boundary: everything of this tick waits for everything of the tick before
        if t > 0:
            for s in all_s:
                for e in ev.values():
                    s.wait_event(e)
        with torch.cuda.stream(sl):
            if t >= 1: stages(S.STAGE_GROUND, (t - 1) % D)
        with torch.cuda.stream(sl2 if variant == "4 streams" else sl):   # (4 streams: the ground fit and the FPS side by side -- what ONE merged launch would do)
            if t >= 3: stages(S.STAGE_FPS, (t - 3) % D)
        with torch.cuda.stream(sa):
            stages(S.STAGE_PROJECT, t % D)
            if t >= 2: stages(S.STAGE_MASK, (t - 2) % D)
        with torch.cuda.stream(sa if variant == "2 streams" else sb):
            if t >= 4: stages(S.STAGE_LABELS | S.STAGE_PLANES | S.STAGE_QUANTISE, (t - 4) % D)
        for s in all_s:
            prev[s].record(s)
        ev = prev
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def run_relaxed(n, two_latency=False):
    """The same roles per stream, but only the true dependencies as events (no tick boundary): L: ground fit(t-1), FPS(t-3);  Ta: pix + band(t), mask(t-2);
    Tb: assign .. quantiser(t-4).  At most one FPS / ground-fit kernel runs at a time (two_latency: one of each)."""
    sl, sa, sb, sl2 = (torch.cuda.Stream(device=dev) for _ in range(4))
    E = lambda: [None] * (n + 8)
    e_band, e_rs, e_mask, e_fps, e_done = E(), E(), E(), E(), E()
    rec = lambda s: (lambda ev: (ev.record(s), ev)[1])(torch.cuda.Event())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(n + 4):
        if t < n:
            with torch.cuda.stream(sa):
                if t >= D and e_done[t - D] is not None: sa.wait_event(e_done[t - D])     # the slot's buffers are free again
                stages(S.STAGE_PROJECT, t % D)
                e_band[t] = rec(sa)
        if 0 <= t - 1 < n:
            with torch.cuda.stream(sl):
                sl.wait_event(e_band[t - 1]); stages(S.STAGE_GROUND, (t - 1) % D); e_rs[t - 1] = rec(sl)
        if 0 <= t - 2 < n:
            with torch.cuda.stream(sa):
                sa.wait_event(e_rs[t - 2]); stages(S.STAGE_MASK, (t - 2) % D); e_mask[t - 2] = rec(sa)
        if 0 <= t - 3 < n:
            lf = sl2 if two_latency else sl
            with torch.cuda.stream(lf):
                lf.wait_event(e_mask[t - 3]); stages(S.STAGE_FPS, (t - 3) % D); e_fps[t - 3] = rec(lf)
        if 0 <= t - 4 < n:
            with torch.cuda.stream(sb):
                sb.wait_event(e_fps[t - 4]); stages(S.STAGE_LABELS | S.STAGE_PLANES | S.STAGE_QUANTISE, (t - 4) % D); e_done[t - 4] = rec(sb)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def run_free(n, depth=3):
    st = [torch.cuda.Stream(device=dev) for _ in range(depth)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(n):
        with torch.cuda.stream(st[t % depth]):
            ops.compress_batch(xyz, offs, tm, gms[t % depth], bufs[t % depth], **kw)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for rep in range(2):
    for v in ("4 streams", "3 streams", "2 streams"):
        run(v, 30)
        dt = run(v, ticks)
        print("tick schedule, %s: %.4f ms per batch, %.0f frames/s" % (v, dt * 1e3, B / dt), flush=True)
    for two in (False, True):
        run_relaxed(30, two)
        dt = run_relaxed(ticks, two)
        print("roles per stream, dependencies only (L%s | Ta | Tb): %.4f ms per batch, %.0f frames/s" % (" | L2" if two else "", dt * 1e3, B / dt), flush=True)
    run_free(30)
    dt = run_free(ticks)
    print("free-running, three batches on three streams (bench.py): %.4f ms per batch, %.0f frames/s" % (dt * 1e3, B / dt), flush=True)
# the schedule computes what the single call computes
ref = ops.BatchBuffers(B, geom, M, dev)
g0 = torch.zeros((B, 4), dtype=torch.float64, device=dev)
ops.compress_batch(xyz, offs, tm, g0, ref, **kw)
torch.cuda.synchronize()
ok = all(torch.equal(ref.seg, b.seg) and torch.equal(ref.nnz, b.nnz) for b in bufs) and all(
    torch.equal(ref.q16[i, :int(ref.nnz[i])], b.q16[i, :int(ref.nnz[i])]) for b in bufs for i in (0, 17, 255))
print("outputs of the scheduled batches equal the single call's:", ok)

"""f4: the input pipeline that keeps the kernels fed -- counterpart of the reference's datalist loop
(tools/compress_datalist.py:91-142,202-206: a thread pool whose workers each load, compress and write one file).

The device part runs at > 10^5 frames/s; a frame is 1.36 MB of points, so the feed is bounded by the host copy into
pinned memory and by the PCIe link (52 GB/s measured = 38.7 k frames/s of 64x2048), not by the kernels.  StreamingCompressor
therefore overlaps the three parts for consecutive batches:

    pool threads   read / copy the frames of batch n+1 straight into a PINNED staging slot (no intermediate concatenation)
    copy stream    slot -> device buffer of batch n+1 (non_blocking H2D), event
    compute stream waits for that event, rpcc_compress_batch + contour codec + payload packing of batch n, D2H of the
                   packed payload into pinned output buffers, event
    pool threads   entropy coding + container + file output of batch n-1 (optional)

submit(n+1) -- with the default depth of 4 also submit(n+2) -- is issued before collect(n), so the copy engine always has a
staged batch waiting.  Slots (staging, device input, BatchBuffers, output) form a ring of `depth`.
"""
import os
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import ops
from .compress_utils import pack_bitstream, pack_frames  # noqa: F401
from .utils import available_cpus


class _Slot:
    def __init__(self, bc, B, cap_points, device, cols=3):
        P, K = bc.T.H * bc.T.W, bc.M + 2
        self.B, self.cap, self.cols = B, int(cap_points), int(cols)    # cols: floats per point in the staging / device input (3 or 4)
        self.xyz_pin = torch.empty((self.cap, self.cols), dtype=torch.float32).pin_memory()
        self.offs_pin = torch.zeros((B + 1,), dtype=torch.int64).pin_memory()
        self.fid_pin = torch.zeros((B,), dtype=torch.int64).pin_memory()
        self.xyz_dev = torch.empty((self.cap, self.cols), dtype=torch.float32, device=device)
        self.offs_dev = torch.zeros((B + 1,), dtype=torch.int64, device=device)
        self.fid_dev = torch.zeros((B,), dtype=torch.int64, device=device)
        self.ground = torch.zeros((B, 4), dtype=torch.float64, device=device)
        self.buf = ops.BatchBuffers(B, bc.T.geom, bc.M, device, max_points=self.cap, general=bc.general)
        self.codec_ws = ops.codec_workspace(B, P, bc.M, device)
        # packed device payload + its pinned mirror: residuals (<= points), index sequence (<= pixels), bits, rows, lengths
        self.qp = torch.empty((self.cap,), dtype=torch.int16, device=device)
        self.sp = torch.empty((B * P,), dtype=torch.int16, device=device)
        self.tot = torch.zeros((2,), dtype=torch.int64, device=device)
        self.qp_pin = torch.empty((self.cap,), dtype=torch.int16).pin_memory()
        self.sp_pin = torch.empty((B * P,), dtype=torch.int16).pin_memory()
        self.bits_pin = torch.empty((B, (P + 7) // 8), dtype=torch.uint8).pin_memory()
        self.model_pin = torch.empty((B, K, 4), dtype=torch.float32).pin_memory()
        self.counts_pin = torch.empty((B, K), dtype=torch.int32).pin_memory()
        self.nnz_pin = torch.empty((B,), dtype=torch.int32).pin_memory()
        self.nseq_pin = torch.empty((B,), dtype=torch.int32).pin_memory()
        self.sal_pin = torch.empty((B, K), dtype=torch.uint8).pin_memory() if bc.general else None
        self.tot_pin = torch.zeros((2,), dtype=torch.int64).pin_memory()
        self.h2d_done = torch.cuda.Event()
        self.done = torch.cuda.Event()
        self.n = 0

    def grow(self, cap_points, device):
        """More points than the slot was sized for (dense sweeps, dual returns: a frame may hold more than one point per
        pixel): the point-count-sized buffers are replaced.  Only called while the slot is idle (nothing of it in flight);
        the projection workspace inside BatchBuffers follows by itself (ops.compress_batch)."""
        self.cap = int(cap_points)
        self.xyz_pin = torch.empty((self.cap, self.cols), dtype=torch.float32).pin_memory()
        self.xyz_dev = torch.empty((self.cap, self.cols), dtype=torch.float32, device=device)
        self.qp = torch.empty((self.cap,), dtype=torch.int16, device=device)
        self.qp_pin = torch.empty((self.cap,), dtype=torch.int16).pin_memory()


class BatchPayload:
    """What the container of every frame of a batch is assembled from: views into a slot's pinned output buffers.
    frame(b) -> the dict compress_point_cloud builds (utils/compress_utils.py:138-179), before the entropy coder."""

    def __init__(self, slot, uniform):
        n = self.n = slot.n
        nnz, nseq = slot.nnz_pin.numpy()[:n].astype(np.int64), slot.nseq_pin.numpy()[:n].astype(np.int64)
        self.qo, self.so = np.concatenate([[0], np.cumsum(nnz)]), np.concatenate([[0], np.cumsum(nseq)])
        self.q16, self.seq = slot.qp_pin.numpy(), slot.sp_pin.numpy().view(np.uint16)
        self.bits, self.model = slot.bits_pin.numpy(), slot.model_pin.numpy()
        counts = slot.counts_pin.numpy()[:n]
        K = counts.shape[1]
        self.nrow = np.where(counts.any(1), K - np.argmax(counts[:, ::-1] != 0, axis=1), 2)   # max(seg)+1 rows (tools/compress.py:102)
        self.sal = slot.sal_pin.numpy() if (slot.sal_pin is not None and not uniform) else None

    def frame(self, b):
        nrow = int(self.nrow[b])
        od = {"residual_quantized": self.q16[self.qo[b]: self.qo[b + 1]]}
        if self.sal is not None:
            od["salience_level"] = self.sal[b, :nrow]
        od["contour_map"] = self.bits[b]
        od["idx_sequence"] = self.seq[self.so[b]: self.so[b + 1]]
        od["plane_param"] = self.model[b, :nrow]
        return od

    def __len__(self):
        return self.n


class StreamingCompressor:
    """bc: pipeline.BatchCompressor (settings: lidar geometry, cluster count, framework, model, entropy back-end).
    batch: frames per device batch; depth: slots in flight; workers: host threads (staging copies, entropy coding)."""

    def __init__(self, bc, batch=256, depth=4, workers=None, points_per_frame=None, pool=None, ingest="xyz"):
        """ingest: "xyz"  -- frames are [N,>=3] arrays; their first three columns are copied into the pinned slot (12 bytes per
                             point staged and sent over the link; a strided host pass when the arrays are .bin rows);
                   "rows" -- the sweeps AS STORED (dataset/dataset.py:48-50: float32 rows x, y, z, intensity): frames are
                             [N,4] arrays (one contiguous copy each) or .bin PATHS, which are read straight into the pinned
                             slot (file -> pinned buffer -> DMA, no pass over the points on the host); the kernels read
                             the rows with a 16-byte stride (rpcc_batch_io.point_stride_bytes = 16).  16 bytes per point
                             cross the link instead of 12."""
        assert ingest in ("xyz", "rows")
        self.ingest = ingest
        self.bc, self.B, self.depth = bc, int(batch), int(depth)
        self.device = bc.device
        P = bc.T.H * bc.T.W
        # initial capacity of a slot in points (default: one per pixel and frame -- a sweep rarely holds more); a batch with
        # more points makes its slot grow (_Slot.grow), so this is a sizing hint, not a limit
        self.cap = int(points_per_frame if points_per_frame is not None else P) * self.B
        self.grown = 0
        self.pool = pool or ThreadPoolExecutor(workers or min(32, available_cpus()))
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.compute_stream = torch.cuda.Stream(device=self.device)
        # On this ROCm build a non_blocking copy_ of several hundred MB from pinned memory still holds the calling host thread
        # until the transfer is over (measured: the call returns when the H2D event fires), so the device work of a batch is
        # issued from a thread of its own and the caller stages the next batch meanwhile.
        self.enq_pool = ThreadPoolExecutor(1)
        self.side_stream = torch.cuda.Stream(device=self.device)     # late copies that must not queue behind the next batch
        self.seq_eager = max(1024, self.B * P // 16)                # index-sequence entries fetched with the batch (typical: P/20 per frame)
        self.slots = [_Slot(bc, self.B, self.cap, self.device, cols=4 if ingest == "rows" else 3) for _ in range(self.depth)]
        self.prof = {"stage": 0.0, "enqueue": 0.0, "collect": 0.0, "drain": 0.0}   # host seconds per phase (diagnostics)
        self.stage_chunks = max(1, min(self.B, 16))      # staging tasks per batch: a few MB each (more, smaller tasks are slower)

    # ---- host staging: frames -> pinned slot (pool threads; numpy releases the GIL inside the copies) -----------------
    def _stage(self, slot, frames, frame_ids):
        n = len(frames)
        assert 0 < n <= self.B
        rows = self.ingest == "rows"
        if rows:
            def rows_of(f):
                if not isinstance(f, (str, os.PathLike)):
                    return f.shape[0]
                nbytes = os.path.getsize(f)
                if nbytes % 16:   # the reference's np.fromfile(...).reshape(-1, 4) raises on such a file (dataset/dataset.py:48-50), and so does ingest="xyz"
                    raise ValueError("%s: %d bytes is no whole number of (x, y, z, intensity) float32 rows" % (f, nbytes))
                return nbytes // 16
            sizes = np.fromiter((rows_of(f) for f in frames), dtype=np.int64, count=n)
        else:
            sizes = np.fromiter((f.shape[0] for f in frames), dtype=np.int64, count=n)
        offs = np.zeros(self.B + 1, np.int64)
        offs[1:n + 1] = np.cumsum(sizes)
        offs[n + 1:] = offs[n]                       # a short last batch: the missing frames are empty
        if offs[n] > slot.cap:      # the slot is idle here (run() drained it): replace its point-sized buffers, with headroom
            slot.grow(int(offs[n]) + int(offs[n]) // 4, self.device)
            self.grown += 1
        dst = slot.xyz_pin.numpy()

        def copy(lo, hi):
            for i in range(lo, hi):
                f = frames[i]
                if not rows:
                    dst[offs[i]:offs[i + 1]] = f[:, :3]   # .bin rows are (x, y, z, intensity): the strided copy drops the 4th column
                elif isinstance(f, (str, os.PathLike)):       # the file's bytes land in the pinned slot: no pass over the points at all
                    with open(f, "rb", buffering=0) as fh:
                        if os.fstat(fh.fileno()).st_size != 16 * int(offs[i + 1] - offs[i]):   # grew or shrank since it was sized
                            raise ValueError("%s changed size while the batch was staged" % f)
                        if offs[i + 1] == offs[i]:
                            continue
                        view = memoryview(dst[offs[i]:offs[i + 1]].reshape(-1).view(np.uint8))
                        got = fh.readinto(view)
                        while 0 < got < len(view):            # (short reads: network file systems)
                            m = fh.readinto(view[got:])
                            if not m:
                                break
                            got += m
                        if got != len(view):
                            raise ValueError("%s changed size while it was read" % f)
                else:
                    assert f.ndim == 2 and f.shape[1] == 4 and f.dtype == np.float32, 'ingest="rows" takes [N,4] float32 rows or .bin paths'
                    dst[offs[i]:offs[i + 1]] = f           # contiguous rows: one memcpy
            return hi - lo
        step = (n + self.stage_chunks - 1) // self.stage_chunks
        list(self.pool.map(lambda lo: copy(lo, min(lo + step, n)), range(0, n, step)))
        slot.offs_pin.numpy()[:] = offs
        fid = np.zeros(self.B, np.int64)
        fid[:n] = np.asarray(frame_ids if frame_ids is not None else np.arange(n), np.int64)[:n]
        slot.fid_pin.numpy()[:] = fid
        slot.n = n
        return int(offs[n])

    # ---- device part: H2D on the copy stream, everything else on the compute stream ------------------------------------
    def _enqueue(self, slot, npts):
        bc = self.bc
        torch.cuda.set_device(self.device)      # (runs on the enqueue thread)
        with torch.cuda.stream(self.copy_stream):
            slot.xyz_dev[:npts].copy_(slot.xyz_pin[:npts], non_blocking=True)
            slot.offs_dev.copy_(slot.offs_pin, non_blocking=True)
            slot.fid_dev.copy_(slot.fid_pin, non_blocking=True)
            slot.h2d_done.record(self.copy_stream)
        with torch.cuda.stream(self.compute_stream):
            self.compute_stream.wait_event(slot.h2d_done)
            nu = None if bc.uniform else ops.nonuniform_cfg(bc.acc, bc.cfg)
            ops.compress_batch(slot.xyz_dev[:npts], slot.offs_dev, bc.T.tm_dev, slot.ground, slot.buf, bc.ground_threshold, bc.acc,
                               ground_seed=bc.seed, frame_ids=slot.fid_dev, model_method=bc.model_method,
                               angle_threshold=bc.cfg.get("plane_angle_threshold", 75), plane_seed=bc.seed, nonuniform=nu)
            bits, seq, nseq = ops.contour_encode(slot.buf.seg, bc.M, ws=slot.codec_ws)
            ops.pack_payload(slot.buf.q16, slot.buf.nnz, packed=slot.qp, capacity=slot.cap, total=slot.tot[0:1])
            ops.pack_payload(seq.view(torch.int16), nseq, packed=slot.sp, capacity=slot.sp.numel(), total=slot.tot[1:2])
            # D2H of what the container needs (upper bounds: the exact lengths are only known on the device)
            slot.tot_pin.copy_(slot.tot, non_blocking=True)
            slot.nnz_pin.copy_(slot.buf.nnz, non_blocking=True)
            slot.nseq_pin.copy_(nseq, non_blocking=True)
            slot.qp_pin[:npts].copy_(slot.qp[:npts], non_blocking=True)         # sum(nnz) <= points of the batch
            slot.sp_pin[:self.seq_eager].copy_(slot.sp[:self.seq_eager], non_blocking=True)   # usually all of it (see _collect)
            slot.bits_pin.copy_(bits, non_blocking=True)
            slot.model_pin.copy_(slot.buf.model, non_blocking=True)
            slot.counts_pin.copy_(slot.buf.counts, non_blocking=True)
            if slot.sal_pin is not None and slot.buf.salience is not None:
                slot.sal_pin.copy_(slot.buf.salience, non_blocking=True)
            slot.keep = (bits, seq, nseq)
            slot.done.record(self.compute_stream)

    # ---- host collection: the batch's payload arrays (views into the pinned buffers) --------------------------------------
    def _collect(self, slot):
        slot.enq.result()                       # the enqueue thread has issued the batch (raises what it raised)
        slot.done.synchronize()
        stot = int(slot.tot_pin[1])
        if stot > self.seq_eager:
            # an unusually long index sequence: the rest is fetched now, on a stream of its own (the compute stream already
            # holds the next batch, which waits for its H2D copy)
            with torch.cuda.stream(self.side_stream):
                slot.sp_pin[self.seq_eager:stot].copy_(slot.sp[self.seq_eager:stot], non_blocking=True)
            self.side_stream.synchronize()
        return BatchPayload(slot, uniform=self.bc.uniform)

    def _encode_chunk(self, payload, lo, hi):
        bc = self.bc
        return pack_frames(bc.bc, [payload.frame(b) for b in range(lo, hi)], uniform=bc.uniform)

    def run(self, batches, sink=None, entropy=True):
        """batches: iterable of (frames, frame_ids) -- frames a list of [N,>=3] float32 arrays (at most `batch` of them),
        frame_ids their identities (or None).  For every batch, in order, sink(index, result) is called with the list of
        .rpcc byte strings (entropy=True) or with the batch's BatchPayload (entropy=False; views into pinned buffers, valid
        inside the sink call only).  The entropy coding of a batch runs on the pool while the next batches are staged and
        computed.  Returns the number of frames processed."""
        submitted = []     # FIFO of (index, slot): on the device, not yet collected
        encoding = []      # FIFO of (index, slot, futures): payloads being entropy-coded from the slot's pinned buffers
        total = 0

        def drain(slot=None):
            # hand finished batches to the sink in order; with `slot` given, until that slot's buffers are free again
            n = 0
            while encoding and (slot is None or any(e[1] is slot for e in encoding)):
                k, _, futs = encoding.pop(0)
                res = [blob for f in futs for blob in f.result()]
                if sink is not None:
                    sink(k, res)
                n += len(res)
            return n

        def finish(item):
            k, slot = item
            payload = self._collect(slot)
            if entropy:   # a few frames per task: the per-frame dictionaries are built on the pool's threads as well
                step = max(1, (payload.n + 4 * self.stage_chunks - 1) // (4 * self.stage_chunks))
                encoding.append((k, slot, [self.pool.submit(self._encode_chunk, payload, lo, min(lo + step, payload.n))
                                           for lo in range(0, payload.n, step)]))
                return 0
            if sink is not None:
                sink(k, payload)
            return payload.n

        for k, (frames, fids) in enumerate(batches):
            slot = self.slots[k % self.depth]
            while any(it[1] is slot for it in submitted):        # (depth 1: the slot's previous batch first)
                total += finish(submitted.pop(0))
            t0 = time.perf_counter()
            total += drain(slot)                                 # its pinned output must not be in use by the entropy coder
            t1 = time.perf_counter()
            npts = self._stage(slot, frames, fids)               # overlaps the device work of the batches in `submitted`
            t2 = time.perf_counter()
            slot.enq = self.enq_pool.submit(self._enqueue, slot, npts)
            t3 = time.perf_counter()
            submitted.append((k, slot))
            while len(submitted) > max(1, self.depth - 2):       # submit(n+1) (and n+2 ...) happened: collect(n)
                total += finish(submitted.pop(0))
            t4 = time.perf_counter()
            self.prof["drain"] += t1 - t0; self.prof["stage"] += t2 - t1; self.prof["enqueue"] += t3 - t2; self.prof["collect"] += t4 - t3
        while submitted:
            total += finish(submitted.pop(0))
        total += drain()
        return total

"""Batched front-end: B frames of one lidar geometry through the fused HIP entry, then the host-side
payload assembly (casts, container, entropy coder) of the reference's compress_point_cloud /
save_compressed_bitstream.  This is the counterpart of the closure body of
tools/compress_datalist.py:91-142 for a whole batch at once."""
import numpy as np
import torch

from . import ops
from .compress_utils import BasicCompressor, pack_bitstream, pack_frames  # noqa: F401


class BatchCompressor:
    SLOTS = 4      # batches in flight one instance supports (submit() without collect()): every one owns its output buffers

    def __init__(self, transformer, cluster_num=100, accuracy=0.02, ground_threshold=0.1, uniform=True,
                 model_method="point", compressor_cfg=None, basic_compressor="bzip2", device=None, seed=0):
        self.M = ops.check_cluster_num(cluster_num)   # (above 254: uint16 labels through the rpcc_*_wide entries; above 65533: refused by name)
        self.T = transformer
        self.device = torch.device(device) if device is not None else transformer.device
        self.acc = accuracy * 2                     # tools/compress.py:46
        self.ground_threshold = ground_threshold
        self.uniform = uniform
        self.model_method = model_method
        self.cfg = compressor_cfg or {}
        self.bc = BasicCompressor(method_name=basic_compressor)
        self.seed = int(seed)
        self._buf = None        # the buffers of the most recent call (tests read them after compress())
        self._codec_ws = None
        self._ring = {}         # frames per batch -> up to SLOTS BatchBuffers (+ codec workspace), busy between submit() and collect()

    @property
    def general(self):
        return not (self.uniform and self.model_method == "point")

    def _buffers(self, B):
        """Output buffers for a batch of B frames: the first set no submit() holds (collect() reads a batch's counts, models, salience and
        contour bits from them, so a second submit() before the first collect() must not land in the same set); at most SLOTS sets per
        batch size, created when first needed.  compress_device() alone never holds a set: its results are valid until the next call."""
        ring = self._ring.setdefault(B, [])
        buf = next((b for b in ring if not b.in_flight), None)
        if buf is None:
            if len(ring) >= self.SLOTS:
                raise RuntimeError("BatchCompressor: %d batches submitted and not collected; collect() one before the next submit()" % len(ring))
            buf = ops.BatchBuffers(B, self.T.geom, self.M, self.device, general=self.general)
            buf.codec_ws = ops.codec_workspace(B, self.T.H * self.T.W, self.M, self.device)
            buf.in_flight = False
            ring.append(buf)
        self._buf, self._codec_ws = buf, buf.codec_ws
        return buf

    def _group_args(self, xyz, offsets, ground=None, frame_ids=None):
        """ops.compress_batch's arguments for this compressor's settings (one geometry group of ops.compress_batch_mixed)."""
        B = offsets.numel() - 1
        buf = self._buffers(B)
        fit = ground is None
        g = torch.zeros((B, 4), dtype=torch.float64, device=self.device) if fit else ground
        # one fused call for all four framework / model combinations (tools/compress.py:109-124)
        nu = None if self.uniform else ops.nonuniform_cfg(self.acc, self.cfg)
        return dict(xyz=xyz, offsets=offsets, tm=self.T.tm_dev, ground=g, buf=buf, ground_seed=self.seed if fit else -1, frame_ids=frame_ids,
                    model_method=self.model_method, angle_threshold=self.cfg.get("plane_angle_threshold", 75), plane_seed=self.seed, nonuniform=nu)

    def _encode(self, args):
        """Contour bits / index sequences of the segmentation a fused call left in args["buf"]."""
        buf = args["buf"]
        sal = None if self.uniform else buf.salience
        bits, seq, nseq = ops.contour_encode(buf.seg, self.M, ws=buf.codec_ws)
        return buf, args["ground"], bits, seq, nseq, sal

    def compress_device(self, xyz, offsets, ground=None, frame_ids=None):
        """Device part.  xyz f32 [sum N,3], offsets i64 [B+1] on the device.  Returns the BatchBuffers plus
        contour bits / index sequences (and salience for the non-uniform framework), all still in HBM.
        frame_ids: stable identities of the frames (utils.frame_identity) for the seeded plane fits."""
        args = self._group_args(xyz, offsets, ground, frame_ids)
        ops.compress_batch(ground_threshold=self.ground_threshold, acc=self.acc, **args)
        return self._encode(args)

    def _upload(self, frames, ground=None):
        """Host arrays of a batch -> device (xyz, offsets, ground or None) on the current stream."""
        offs = np.zeros(len(frames) + 1, np.int64)
        offs[1:] = np.cumsum([f.shape[0] for f in frames])
        xyz = torch.from_numpy(np.ascontiguousarray(np.concatenate([f[:, :3] for f in frames]), dtype=np.float32)).to(self.device)
        gnd = None if ground is None else torch.from_numpy(np.asarray(ground, np.float64).reshape(-1, 4)).to(self.device)
        return xyz, torch.from_numpy(offs).to(self.device), gnd, int(offs[-1])

    def _payload(self, n, npts, xyz, dev_out):
        """The two variable-length 16-bit streams leave the device with the frames back to back (rpcc_pack_payload), not as the padded
        [B,P] arrays: nnz <= points of the frame, one index per contour start <= pixels.  -> the context collect() takes."""
        buf, g, bits, seq, nseq, sal = dev_out
        qp, qtot = ops.pack_payload(buf.q16, buf.nnz, capacity=npts)
        sp, stot = ops.pack_payload(seq.view(torch.int16), nseq)
        buf.in_flight = True    # until collect() has read it (_buffers)
        return dict(n=n, buf=buf, bits=bits, nseq=nseq, sal=sal, qp=qp, qtot=qtot, sp=sp, stot=stot,
                    stream=torch.cuda.current_stream(self.device), keep=(xyz, g, seq))

    def submit(self, frames, ground=None, frame_ids=None):
        """Device part of compress() on the current stream, nothing waited for.  -> a context for collect()."""
        xyz, offs, gnd, npts = self._upload(frames, ground)
        return self._payload(len(frames), npts, xyz, self.compress_device(xyz, offs, gnd, frame_ids))

    def collect(self, ctx, pool=None):
        """Waits for submit()'s stream and assembles the .rpcc byte strings (host part: casts, container, entropy coder).
        pool: a concurrent.futures executor -- the frames' entropy coding then runs on its threads (bz2 / zlib / lz4
        release the GIL), like the reference's --workers ThreadPoolExecutor (tools/compress_datalist.py:202-206)."""
        buf = ctx["buf"]
        try:
            ctx["stream"].synchronize()
            bits, nseq, sal = ctx["bits"], ctx["nseq"], ctx["sal"]
            nnz, nseq_h = buf.nnz.cpu().numpy(), nseq.cpu().numpy()
            seg_max = buf.counts.cpu().numpy()
            q16 = ctx["qp"][: int(ctx["qtot"].item())].cpu().numpy()
            seq_h = ctx["sp"][: int(ctx["stot"].item())].cpu().numpy().view(np.uint16)
            qo, so = np.concatenate([[0], np.cumsum(nnz)]), np.concatenate([[0], np.cumsum(nseq_h)])
            bits_h, model = bits.cpu().numpy(), buf.model.cpu().numpy()
            sal_h = None if sal is None else sal.cpu().numpy()
        finally:
            buf.in_flight = False   # everything collect() needs is on the host -- or the call failed: either way the slot is free again
        def assemble(b):
            nrow = int(np.flatnonzero(seg_max[b])[-1]) + 1          # max(seg)+1 rows (tools/compress.py:102)
            od = {"residual_quantized": q16[qo[b]: qo[b + 1]]}
            if sal_h is not None:
                od["salience_level"] = sal_h[b, :nrow]
            od["contour_map"] = bits_h[b]
            od["idx_sequence"] = seq_h[so[b]: so[b + 1]]
            od["plane_param"] = model[b, :nrow]
            return od
        # a few frames per task: one call into the host library per chunk (compress_utils.pack_frames)
        n = ctx["n"]
        step = max(1, n // 64) if pool is not None else max(n, 1)
        chunk = lambda lo: pack_frames(self.bc, [assemble(b) for b in range(lo, min(lo + step, n))], uniform=self.uniform)
        parts = list(pool.map(chunk, range(0, n, step))) if pool is not None else [chunk(lo) for lo in range(0, n, step)]
        return [blob for part in parts for blob in part]

    def discard(self, ctx):
        """Gives up a submit() whose results are not wanted (a caller that aborts a batch): waits for its stream -- the kernels may still
        be writing the slot -- and frees the slot.  Without it (or collect()) the slot stays busy and the ring runs out after SLOTS such calls."""
        try:
            ctx["stream"].synchronize()
        finally:
            ctx["buf"].in_flight = False

    def compress(self, frames, ground=None, pool=None, frame_ids=None):
        """frames: list of [N,3] arrays.  -> list of .rpcc byte strings (one per frame)."""
        return self.collect(self.submit(frames, ground, frame_ids), pool=pool)


class MixedBatchCompressor:
    """BASELINE configs[4]: one batch holding sweeps of several lidar geometries (variable H x W).  The frames are grouped by
    geometry and the groups go through ONE fused call (ops.compress_batch_mixed / rpcc_compress_batch_mixed): the kernels with one
    workgroup per frame or per label -- ground RANSAC, FPS, plane fits: latency-bound whatever the image size -- run once over all
    groups, the pixel-parallel ones group after group; the results come back in input order.  (Until round 4 every group ran as its
    own chain of launches on its own stream: the device runs three or four kernels side by side, so three chains of small
    launches left most of it idle -- profiles/HISTORY.md.)
    transformers: {lidar name: PCTransformer}; the other arguments are BatchCompressor's."""

    SLOTS = BatchCompressor.SLOTS   # mixed batches in flight of the streaming form (submit / collect on SLOTS streams): every lidar's compressor
                                    # keeps that many buffer sets, a further submit() before a collect() raises

    def __init__(self, transformers, **kw):
        self.bcs = {name: BatchCompressor(t, **kw) for name, t in transformers.items()}
        first = next(iter(self.bcs.values()))
        self.device, self.ground_threshold, self.acc = first.device, first.ground_threshold, first.acc

    def compress_device(self, parts):
        """parts: {lidar name: (xyz, offsets, ground or None, frame_ids or None)} on the device.  One fused call on the current stream
        -> {lidar name: what BatchCompressor.compress_device returns}."""
        args = {name: self.bcs[name]._group_args(*part) for name, part in parts.items()}
        ops.compress_batch_mixed(list(args.values()), ground_threshold=self.ground_threshold, acc=self.acc)
        return {name: self.bcs[name]._encode(a) for name, a in args.items()}

    def submit(self, frames, lidars):
        """Device part of compress() on the current stream, nothing waited for.  -> a context for collect()."""
        groups = {}
        for i, name in enumerate(lidars):
            groups.setdefault(name, []).append(i)
        up = {name: self.bcs[name]._upload([frames[i] for i in idx]) for name, idx in groups.items()}
        outs = self.compress_device({name: (xyz, offs, gnd, None) for name, (xyz, offs, gnd, _) in up.items()})
        ctxs = {name: self.bcs[name]._payload(len(groups[name]), up[name][3], up[name][0], outs[name]) for name in groups}
        return dict(n=len(frames), groups=groups, ctxs=ctxs)

    def collect(self, ctx, pool=None):
        out = [None] * ctx["n"]
        for name, idx in ctx["groups"].items():
            for i, blob in zip(idx, self.bcs[name].collect(ctx["ctxs"][name], pool=pool)):
                out[i] = blob
        return out

    def compress(self, frames, lidars):
        """frames: list of [N,3] arrays; lidars: the lidar name of every frame.  -> list of .rpcc byte strings."""
        return self.collect(self.submit(frames, lidars))

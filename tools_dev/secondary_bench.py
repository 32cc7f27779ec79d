#!/usr/bin/env python3
"""Developer benchmark for the rows next to the headline path (configs[2]/[4], f1-f3): non-uniform framework, plane
model, contour codec, payload packing, decoder -- B frames of 64x2048, each stage alone, wall clock per stage.
Under `rocprofv3 --kernel-trace --stats` this is the command behind profiles/r01_secondary_kernel_stats.md."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rpcc_amd  # noqa: F401
from rpcc_amd import ops, synth
from rpcc_amd.transformer import PCTransformer
from rpcc_amd.pipeline import BatchCompressor

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
cfg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "r-pcc_amd", "lidar_cfg", "Velodyne_HDL_64E_2048.yaml")
T = PCTransformer(cfg)
xyz, offs = synth.make_batch(range(B), T.H, T.W, device=dev)


def timed(name, fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
    torch.cuda.synchronize()
    print("%-46s %8.3f ms per %d frames" % (name, (time.perf_counter() - t0) / N * 1e3, B), flush=True)


for uniform in (True, False):
    for mm in ("point", "plane"):
        bc = BatchCompressor(T, uniform=uniform, model_method=mm, device=dev)
        timed("compress_device uniform=%s model=%s" % (uniform, mm), lambda: bc.compress_device(xyz, offs))

bc = BatchCompressor(T, uniform=False, model_method="plane", device=dev)
buf, g, bits, seq, nseq, sal = bc.compress_device(xyz, offs)
torch.cuda.synchronize()
M, H, W = bc.M, T.H, T.W
cws = ops.codec_workspace(B, H * W, M, dev)
timed("contour_encode", lambda: ops.contour_encode(buf.seg, M, ws=cws))
timed("contour_decode", lambda: ops.contour_decode(bits, seq, H, W, M, ws=cws))
packed = torch.zeros((int(offs[-1].item()),), dtype=torch.int16, device=dev)
tot = torch.zeros((1,), dtype=torch.int64, device=dev)
timed("pack_payload (residual stream)", lambda: ops.pack_payload(buf.q16, buf.nnz, packed=packed, capacity=packed.numel(), total=tot))
la = (np.array([bc.acc] * 4) + np.array((0, 0.02, 0.04, 0.06))).astype(np.float32)
timed("decode (non-uniform steps, with points)", lambda: ops.decode(buf.seg, buf.q16, buf.model, T.tm_dev, la, salience=sal, want_points=True, ws=cws))

# three calls in flight (own buffers, own streams), like bench.py does for the headline configuration
for uniform, mm in ((True, "plane"), (False, "point"), (False, "plane")):
    bcs = [BatchCompressor(T, uniform=uniform, model_method=mm, device=dev) for _ in range(3)]
    sts = [torch.cuda.Stream(device=dev) for _ in range(3)]
    for bcx, st in zip(bcs, sts):
        with torch.cuda.stream(st):
            bcx.compress_device(xyz, offs)
    torch.cuda.synchronize()
    R = 12
    t0 = time.perf_counter()
    for i in range(R):
        with torch.cuda.stream(sts[i % 3]):
            bcs[i % 3].compress_device(xyz, offs)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / R
    print("3 in flight: uniform=%s model=%s  %.3f ms per %d frames (%.0f frames/s)" % (uniform, mm, dt * 1e3, B, B / dt), flush=True)

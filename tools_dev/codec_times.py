"""GPU: per-call times of the stages beside the fused batch call -- contour codec, payload packing, decoder, key points -- on 256 synthetic 64 x 2048
sweeps (events around 20 repetitions each; everything resident).  A sanity sweep for paths no headline shows.
Usage: python tools_dev/codec_times.py"""
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
hf, vmax, vmin = 2 * np.pi, 2.0 * np.pi / 180, -24.9 * np.pi / 180
geom = ops.make_geom(H, W, hf, vmax, vmin)
tm = torch.from_numpy(ops.transform_map(H, W, hf, vmax, vmin)).to(dev)
ids = list(range(20000, 20000 + B))
xyz, offs = synth.make_batch(ids, H, W, device=dev)
buf = ops.BatchBuffers(B, geom, M, dev, general=True)
gms = torch.zeros((B, 4), dtype=torch.float64, device=dev)
fid = torch.as_tensor(np.asarray(ids, np.int64), device=dev)


def timed(name, fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record(); torch.cuda.synchronize()
    print("%-58s %9.1f us per call of %d frames" % (name, e0.elapsed_time(e1) * 1e3 / reps, B), flush=True)
    return out


timed("compress_batch (uniform + point)", lambda: ops.compress_batch(xyz, offs, tm, gms, buf, ground_seed=1, frame_ids=fid))
nu = ops.nonuniform_cfg(0.04)
timed("compress_batch (non-uniform + plane)", lambda: ops.compress_batch(xyz, offs, tm, gms, buf, ground_seed=1, frame_ids=fid, model_method="plane", plane_seed=1, nonuniform=nu))
ops.compress_batch(xyz, offs, tm, gms, buf, ground_seed=1, frame_ids=fid)
torch.cuda.synchronize()
cws = ops.codec_workspace(B, H * W, M, dev)
enc = timed("contour_encode", lambda: ops.contour_encode(buf.seg, M, ws=cws))
bits, seq, nseq = enc[0], enc[1], enc[2]
timed("contour_decode", lambda: ops.contour_decode(bits, seq, H, W, M, ws=cws))
timed("pack_payload", lambda: ops.pack_payload(buf.q16, buf.nnz))
timed("decode (range image)", lambda: ops.decode(buf.seg, buf.q16, buf.model, tm, 0.04, ws=cws))
timed("decode (range image + points)", lambda: ops.decode(buf.seg, buf.q16, buf.model, tm, 0.04, want_points=True, ws=cws))
timed("extract_features", lambda: ops.extract_features(buf.ri, buf.seg))
timed("backproject", lambda: ops.backproject(buf.ri, tm))
timed("point_model", lambda: ops.point_model(buf.ri, buf.seg, gms, M))
timed("predict_quantize", lambda: ops.predict_quantize(buf.ri, tm, buf.seg, buf.model, 0.04, M, int16=True))
timed("intra_predict", lambda: ops.intra_predict(buf.seg, buf.model, tm))
timed("project (stand-alone entry)", lambda: ops.project(xyz, offs, geom))
timed("ground_ransac", lambda: ops.ground_ransac(buf.ri, tm, seed=1, frame_ids=fid))
timed("assign (stand-alone entry)", lambda: ops.assign(buf.ri, tm, gms, buf.centers))
pts = [xyz[int(offs[i]):int(offs[i + 1])] for i in range(4)]
n = (min(p.shape[0] for p in pts) // 4) * 4
pl = torch.stack([p[:n] for p in pts]).contiguous()
timed("fps_xyz: 4 point lists of %d points, 100 samples" % n, lambda: ops.fps_xyz(pl, 100))
timed("fps_xyz: 1 point list", lambda: ops.fps_xyz(pl[:1].contiguous(), 100))
timed("fps_xyz brute force: 1 point list", lambda: ops.fps_xyz(pl[:1].contiguous(), 100, bruteforce=True))
timed("fps_xyz: 1 point list, N % 4 != 0 (scalar loads)", lambda: ops.fps_xyz(pl[:1, :n - 1].contiguous(), 100))
# the same clouds with consecutive points neighbours in space: the pixels of the range image in row-major order
pc = ops.backproject(buf.ri, tm).reshape(B, -1, 3)
keep = [pc[i][buf.ri[i].reshape(-1) != 0] for i in range(4)]
n2 = (min(k.shape[0] for k in keep) // 4) * 4
cl = torch.stack([k[:n2] for k in keep]).contiguous()
print("probe:", ops.fps_xyz_probe(pl).tolist(), ops.fps_xyz_probe(cl).tolist())
timed("fps_xyz: 4 row-major lists of %d points" % n2, lambda: ops.fps_xyz(cl, 100))
timed("fps_xyz: 1 row-major list", lambda: ops.fps_xyz(cl[:1].contiguous(), 100))
timed("fps_xyz brute force: 1 row-major list", lambda: ops.fps_xyz(cl[:1].contiguous(), 100, bruteforce=True))

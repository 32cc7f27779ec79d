import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import rpcc_amd
from rpcc_amd import ops, synth
dev = torch.device("cuda:0")
for (H, W, vmax, vmin, name) in ((64, 2048, 2.0, -24.9, "64x2048"), (16, 1800, 15.0, -15.0, "16x1800")):
    geom = ops.make_geom(H, W, 2 * np.pi, vmax * np.pi / 180, vmin * np.pi / 180)
    tm = torch.from_numpy(ops.transform_map(H, W, 2 * np.pi, vmax * np.pi / 180, vmin * np.pi / 180)).to(dev)
    B = 256
    frames = []
    for i in range(8):
        f = synth.make_frame(100 + i, H, W, vmax_deg=vmax, vmin_deg=vmin).numpy()
        frames.append(f[f[:, 2] > -1.45])          # no ground candidate at all: the fit runs on every pixel
    fr = [frames[i % 8] for i in range(B)]
    offs = np.zeros(B + 1, np.int64); offs[1:] = np.cumsum([f.shape[0] for f in fr])
    xyz = torch.from_numpy(np.concatenate(fr)).to(dev); o = torch.from_numpy(offs).to(dev)
    ri = ops.project(xyz, o, geom)
    fid = torch.arange(B, device=dev, dtype=torch.int64)
    for _ in range(3): ops.ground_ransac(ri, tm, seed=1, frame_ids=fid)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): ops.ground_ransac(ri, tm, seed=1, frame_ids=fid)
    torch.cuda.synchronize()
    print("%s: ground fit on the whole cloud, 256 frames: %.1f us per launch" % (name, (time.perf_counter() - t0) / 10 * 1e6))

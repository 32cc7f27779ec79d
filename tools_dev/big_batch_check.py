import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, time
import rpcc_amd
from rpcc_amd import ops, synth
from oracle import oracle as orc
dev = torch.device("cuda:0")
gd = orc.GEOMS["VelodyneVLP16"]; g = orc.LidarGeom(**gd); tm = orc.transform_map(g)
geom = ops.make_geom(g.H, g.W, g.horizontal_FOV, g.vertical_max, g.vertical_min)
NB = 64
base = [synth.make_frame(50 + i, g.H, g.W, vmax_deg=gd["vmax_deg"], vmin_deg=gd["vmin_deg"]).numpy() for i in range(NB)]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
frames = [base[i % NB] for i in range(B)]
offs = np.zeros(B + 1, np.int64); offs[1:] = np.cumsum([f.shape[0] for f in frames])
xyz = torch.from_numpy(np.concatenate(frames)).to(dev)
buf = ops.BatchBuffers(B, geom, 100, dev, max_points=xyz.shape[0])
gfit = torch.zeros((B, 4), dtype=torch.float64, device=dev)
tmd = torch.from_numpy(tm).to(dev); od = torch.from_numpy(offs).to(dev)
ops.compress_batch(xyz, od, tmd, gfit, buf, ground_seed=5); torch.cuda.synchronize()
t0 = time.perf_counter(); ops.compress_batch(xyz, od, tmd, gfit, buf, ground_seed=5); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("B=%d VLP16: %.2f ms, %.0f frames/s, ws %.1f GB" % (B, dt * 1e3, B / dt, buf.ws.numel() / 1e9))
bad = 0
for i in [0, 1, NB, B // 2, B - 2, B - 1, 12345 % B, 777 % B]:
    f = frames[i]; r = orc.project(f, g); gm = orc.ground_model(r, tm, seed=5 + i); e = orc.compress_frame(f, g, tm, gm)
    n = int(buf.nnz[i])
    ok = (np.array_equal(buf.ri[i].cpu().numpy().view(np.uint32), r.view(np.uint32)) and np.array_equal(buf.seg[i].cpu().numpy(), e["seg_idx"].astype(np.uint8))
          and n == e["q"].shape[0] and np.array_equal(buf.q16[i, :n].cpu().numpy(), e["q"].astype(np.int16)))
    bad += not ok
print("mismatching of 8 sampled frames:", bad)

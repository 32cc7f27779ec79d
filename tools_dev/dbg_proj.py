import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import rpcc_amd
from rpcc_amd import ops
from oracle import oracle as orc
g = orc.LidarGeom(**orc.GEOMS["Velodyne64E"]); geom = ops.make_geom(g.H, g.W, g.horizontal_FOV, g.vertical_max, g.vertical_min)
rng = np.random.default_rng(21)
a = rng.normal(0, 20, (60000, 3)).astype(np.float32)
a[:, 2] = rng.normal(-1, 1.5, 60000)
e = a[:1000].copy()
e[5] = [np.nan, 1, 1]; e[6] = [np.inf, 1, 1]; e[7] = [1e30, 1e30, 0]
keep = np.ones(1000, bool); keep[[5, 6, 7]] = False
dev = torch.device("cuda:0")
for atomic in (False, True):
    ri = ops.project(torch.from_numpy(e).to(dev), torch.tensor([0, 1000], dtype=torch.int64, device=dev), geom, atomic_path=atomic).cpu().numpy()[0]
    ref = orc.project(e[keep], g)
    d = np.argwhere(ri.view(np.uint32) != ref.view(np.uint32))
    print(atomic, len(d), d[:5], [(ri[tuple(i)], ref[tuple(i)]) for i in d[:5]])
print(ops.project_fastpath_check(torch.from_numpy(e).to(dev), geom))

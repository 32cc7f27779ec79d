"""GPU box, library built with -DRPCC_DEVTRACE: cycles per phase of project_ordered_kernel's chunk loop (workgroup 0, thread 0) on the real sweep in stored
order, B frames.  usage: (RPCC_EXTRA_FLAGS=-DRPCC_DEVTRACE python r-pcc_amd/build.py) python tools_dev/ordered_phases.py [B]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rpcc_amd  # noqa: E402,F401
from rpcc_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H, W = 64, 2000
hf, vmax, vmin = 2 * np.pi, 2.0 * np.pi / 180, -24.9 * np.pi / 180
geom = ops.make_geom(H, W, hf, vmax, vmin)
xyz1 = np.load(os.path.join(ROOT, "tests", "golden", "example_64E.npz"))["xyz"]
xyz = torch.from_numpy(np.tile(xyz1, (B, 1))).to(dev)
offs = torch.arange(B + 1, dtype=torch.int64, device=dev) * xyz1.shape[0]
acc = torch.zeros(B, dtype=torch.int32, device=dev)
for _ in range(3):
    ops.project(xyz, offs, geom, accepted=acc)
torch.cuda.synchronize()
stamps = torch.zeros(4096, dtype=torch.int64, device=dev)
_lib.check(_lib.lib().rpcc_debug_stamps(_lib.ptr(stamps)))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ops.project(xyz, offs, geom, accepted=acc)
e1.record()
torch.cuda.synchronize()
_lib.lib().rpcc_debug_stamps(None)
s = stamps.cpu().numpy()[900:916]
names = ["chunk end barrier", "compute + window atomics", "next loads + pending notes", "barrier 1", "exact drain", "barrier 2", "window moves",
         "(tail)", "final write-out"]
print("accepted %d of %d; whole projection %.1f us (events)" % (int(acc.sum()), B, e0.elapsed_time(e1) * 1e3))
tot = s[:9].sum()
for i, nme in enumerate(names):
    print("   %-28s %9d cycles  %5.1f %%" % (nme, s[i], 100.0 * s[i] / max(tot, 1)))
print("   chunks %d, window moves %d, total %d cycles = %.1f us at 2.4 GHz" % (s[15], s[14], tot, tot / 2400.0))

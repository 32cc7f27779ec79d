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
    ops.project(xyz, offs, geom, accepted=acc, order_flags=ops.PROJECT_ORDER_PROBE)
torch.cuda.synchronize()
stamps = torch.zeros(4096, dtype=torch.int64, device=dev)
_lib.check(_lib.lib().rpcc_debug_stamps(_lib.ptr(stamps)))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ops.project(xyz, offs, geom, accepted=acc, order_flags=ops.PROJECT_ORDER_PROBE)
e1.record()
torch.cuda.synchronize()
_lib.lib().rpcc_debug_stamps(None)
s = stamps.cpu().numpy()[900:916]
names = {0: "chunk top", 4: "exact sequence (previous chunk's queue)", 1: "compute + window atomics", 2: "next loads + pending notes", 3: "barrier 1",
         6: "window moves", 7: "(tail)", 8: "final write-out + zero rows", 9: "hand-off pass (candidate bytes)"}
print("accepted %d of %d; whole projection %.1f us (events)" % (int(acc.sum()), B, e0.elapsed_time(e1) * 1e3))
tot = s[:10].sum()
for i in (0, 4, 1, 2, 3, 6, 7, 8, 9):
    print("   %-40s %9d cycles  %5.1f %%" % (names[i], s[i], 100.0 * s[i] / max(tot, 1)))
print("   chunks %d, window moves %d, total %d cycles = %.1f us at 2.4 GHz" % (s[15], s[14], tot, tot / 2400.0))

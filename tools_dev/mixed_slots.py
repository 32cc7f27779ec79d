"""GPU box: configs[4] on one GPU by the number of mixed batches in flight (bench.run_mixed).  usage: python tools_dev/mixed_slots.py"""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpcc_amd  # noqa: E402,F401
import bench  # noqa: E402

a = types.SimpleNamespace(accuracy=0.02)
ctx = dict(dev=torch.device("cuda:0"))
print("| frames per geometry and mixed batch | mixed batches in flight | frames/s | ms per mixed batch | verified |")
print("|---|---|---|---|---|")
for per in (85, 256):
    for slots in (1, 2, 3, 4):
        for rep in range(2):
            r = bench.run_mixed(a, ctx, per=per, reps=24, slots=slots)
            print("| %d | %d | %.0f | %.3f | %s |" % (per, slots, r["value"], r["ms_per_mixed_batch"], r["verified"]), flush=True)

"""GPU box: configs[4] on one GPU -- one fused call per mixed batch (rpcc_compress_batch_mixed) against the groups as chains of launches on
their own streams, by mixed batches in flight.  usage: python tools_dev/mixed_ab.py [frames per geometry ...]"""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpcc_amd  # noqa: E402,F401
import bench  # noqa: E402

a = types.SimpleNamespace(accuracy=0.02)
ctx = dict(dev=torch.device("cuda:0"))
print("| frames per geometry and mixed batch | form | mixed batches in flight | frames/s | ms per mixed batch | verified |")
print("|---|---|---|---|---|---|")
for per in [int(v) for v in sys.argv[1:]] or [85, 256]:
    for by_streams in (True, False):
        for slots in (1, 2, 3, 4):
            for rep in range(2):
                r = bench.run_mixed(a, ctx, per=per, reps=24, slots=slots, by_streams=by_streams)
                print("| %d | %s | %d | %.0f | %.3f | %s |" % (per, "groups on streams" if by_streams else "one fused call", slots, r["value"],
                                                               r["ms_per_mixed_batch"], r["verified"]), flush=True)

#!/usr/bin/env python3
"""Developer benchmark (GPU box): the datalist feed FROM FILES -- 256 synthetic 64x2048 sweeps written as KITTI-style .bin files
(float32 rows x, y, z, intensity) into /dev/shm -- through loader.StreamingCompressor without the entropy coder:
  ingest="xyz"   what the reference's loader does (dataset/dataset.py:48-50,62): np.fromfile -> reshape(-1, 4) -> [:, :3], then the
                 copy into the pinned slot (two passes over the points on the host, 12 B per point over the link)
  ingest="rows"  the file's bytes are read straight into the pinned slot and go to the device as stored (no pass over the points
                 besides the read itself, 16 B per point over the link, 16-byte row loads in the pixel kernel)
Prints frames/s, host CPU seconds per frame (process time over all threads) and the host milliseconds per batch by phase.
usage: python3 tools_dev/loader_files_bench.py [batches] [workers]"""
import os
import shutil
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpcc_amd  # noqa: E402,F401
from rpcc_amd import synth  # noqa: E402
from rpcc_amd.dataset import DatasetTemplate  # noqa: E402
from rpcc_amd.loader import StreamingCompressor  # noqa: E402
from rpcc_amd.pipeline import BatchCompressor  # noqa: E402
from rpcc_amd.transformer import PCTransformer  # noqa: E402

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 16
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 16
B = 256
d = tempfile.mkdtemp(prefix="rpcc_bins_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    paths = []
    for i in range(B):
        f = synth.make_frame(i, 64, 2048, device="cuda:0").cpu().numpy()
        p = os.path.join(d, "%06d.bin" % i)
        np.concatenate([f, np.full((f.shape[0], 1), 0.5, np.float32)], 1).tofile(p)
        paths.append(p)
    T = PCTransformer(dict(HORIZONTAL_FOV=360, VERTICAL_ANGLE_MAX=2.0, VERTICAL_ANGLE_MIN=-24.9, RANGE_IMAGE_HEIGHT=64, RANGE_IMAGE_WIDTH=2048))
    bc = BatchCompressor(T, seed=1)
    pool = ThreadPoolExecutor(workers)
    res = {}
    print("| ingest | frames/s | host CPU ms per frame | host ms per batch: read files / stage / enqueue / collect |")
    print("|---|---|---|---|")
    for rep in range(2):
        for ingest in ("xyz", "rows"):
            sc = StreamingCompressor(bc, batch=B, depth=4, workers=workers, pool=pool, ingest=ingest)
            t_read = [0.0]

            def batches(n):
                for k in range(n):
                    t0 = time.perf_counter()
                    frames = paths if ingest == "rows" else list(pool.map(DatasetTemplate.load_data, paths))
                    t_read[0] += time.perf_counter() - t0
                    yield frames, list(range(k * B, k * B + B))
            got = {}
            sc.run(batches(2), sink=lambda k, r: got.setdefault(ingest, [np.array(r.frame(b)["residual_quantized"], copy=True) for b in range(4)]) if k == 0 else None, entropy=False)
            res[ingest] = got[ingest]
            for k in sc.prof:
                sc.prof[k] = 0.0
            t_read[0] = 0.0
            t0, c0 = time.perf_counter(), time.process_time()
            n = sc.run(batches(NB), entropy=False)
            torch.cuda.synchronize()
            dt, cpu = time.perf_counter() - t0, time.process_time() - c0
            print("| %s | %.0f | %.3f | %.2f / %.2f / %.2f / %.2f |" % (ingest, n / dt, cpu / n * 1e3, t_read[0] / NB * 1e3, sc.prof["stage"] / NB * 1e3,
                                                                   sc.prof["enqueue"] / NB * 1e3, sc.prof["collect"] / NB * 1e3), flush=True)
    print("payloads of the two ingests identical:", all(np.array_equal(a, b) for a, b in zip(res["xyz"], res["rows"])))
finally:
    shutil.rmtree(d, ignore_errors=True)

#!/usr/bin/env python3
"""Developer benchmark: BatchCompressor.compress_device for the four framework / model combinations (configs[2], [4])."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rpcc_amd
from rpcc_amd import synth
from rpcc_amd.transformer import PCTransformer
from rpcc_amd.pipeline import BatchCompressor

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
cfg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "r-pcc_amd", "lidar_cfg", "Velodyne_HDL_64E_2048.yaml")
T = PCTransformer(cfg, device=dev) if "device" in PCTransformer.__init__.__code__.co_varnames else PCTransformer(cfg)
xyz, offs = synth.make_batch(range(B), T.H, T.W, device=dev)
for uniform in (True, False):
    for mm in ("point", "plane"):
        bc = BatchCompressor(T, uniform=uniform, model_method=mm, device=dev)
        bc.compress_device(xyz, offs); torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            bc.compress_device(xyz, offs)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print("uniform=%-5s model=%-5s  %.3f ms per %d frames  (%.0f frames/s)" % (uniform, mm, dt * 1e3, B, B / dt))

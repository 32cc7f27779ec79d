#!/usr/bin/env python3
"""Developer experiment: single-call latency of the fused entry, direct launches vs one captured HIP graph (torch.cuda.CUDAGraph)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rpcc_amd  # noqa: F401
from rpcc_amd import ops, synth
dev = torch.device("cuda:0")
H, W, M = 64, 2048, 100
hfov, vmax, vmin = 2 * np.pi, 2.0 * np.pi / 180, -24.9 * np.pi / 180
geom = ops.make_geom(H, W, hfov, vmax, vmin)
tm = torch.from_numpy(ops.transform_map(H, W, hfov, vmax, vmin)).to(dev)
for B in (1, 4, 16):
    xyz, offs = synth.make_batch(range(B), H, W, device=dev)
    buf = ops.BatchBuffers(B, geom, M, dev, max_points=xyz.shape[0])
    g = torch.zeros((B, 4), dtype=torch.float64, device=dev)
    run = lambda: ops.compress_batch(xyz, offs, tm, g, buf, ground_seed=0)
    for _ in range(3): run()
    torch.cuda.synchronize()
    ref = (buf.seg.clone(), buf.q16.clone(), buf.nnz.clone())
    n = 200
    t0 = time.perf_counter()
    for _ in range(n): run(); torch.cuda.synchronize()
    t_direct = (time.perf_counter() - t0) / n
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        run()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        run()
    buf.seg.zero_(); buf.nnz.zero_()
    graph.replay(); torch.cuda.synchronize()
    nn = int(ref[2].max())
    ok = torch.equal(buf.seg, ref[0]) and torch.equal(buf.nnz, ref[2]) and torch.equal(buf.q16[:, :nn // 2], ref[1][:, :nn // 2])
    t0 = time.perf_counter()
    for _ in range(n): graph.replay(); torch.cuda.synchronize()
    t_graph = (time.perf_counter() - t0) / n
    print("B=%d: direct %.3f ms, graph replay %.3f ms, identical outputs: %s" % (B, t_direct * 1e3, t_graph * 1e3, ok))

"""GPU box: the headline step with the pipeline slots' streams confined to CU subsets (hipExtStreamCreateWithCUMask) -- VERDICT round 3,
item 2a "placement, not priorities".  Modes: none (the bench's plain streams), disjoint (slot k owns every third CU), xcd (slot k owns
XCDs {k, k+3, k+6}: 96 CUs; the 8th and 9th XCD-slots fold), two_thirds (slot k is kept OFF one third of the CUs), half_lat (one extra
stream per slot on half of the CUs is not possible without forking inside the library: not measured here).
usage: python tools_dev/cu_mask_bench.py [steps]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpcc_amd  # noqa: E402,F401
from rpcc_amd import ops, synth  # noqa: E402

hip = C.CDLL("libamdhip64.so")
NCU, DEPTH, B, H, W, M = 256, 3, 256, 64, 2048, 100


def masked_stream(cus):
    words = (C.c_uint32 * (NCU // 32))()
    for c in cus:
        words[c // 32] |= 1 << (c % 32)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(NCU // 32), words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


def masks(mode):
    # CU index -> (XCD, CU in XCD) is not documented: both interleavings are tried (cu % 8 = XCD, and cu // 32 = XCD)
    if mode == "none":
        return None
    if mode == "disjoint":
        return [[c for c in range(NCU) if c % 3 == k] for k in range(DEPTH)]
    if mode == "two_thirds":
        return [[c for c in range(NCU) if c % 3 != k] for k in range(DEPTH)]
    if mode == "xcd_mod":
        return [[c for c in range(NCU) if (c % 8) % 3 == k] for k in range(DEPTH)]
    if mode == "xcd_div":
        return [[c for c in range(NCU) if (c // 32) % 3 == k] for k in range(DEPTH)]
    if mode == "half":
        return [[c for c in range(NCU) if (c + k) % 2 == 0 or k == 2] for k in range(DEPTH)]
    raise SystemExit(mode)


def run(mode, steps):
    dev = torch.device("cuda:0")
    hfov, vmax, vmin = 2 * np.pi, np.radians(2.0), np.radians(-24.9)
    geom = ops.make_geom(H, W, hfov, vmax, vmin)
    tm = torch.from_numpy(ops.transform_map(H, W, hfov, vmax, vmin)).to(dev)
    xyz, offs = synth.make_batch(list(range(B)), H, W, device=dev)
    fid = torch.arange(B, dtype=torch.int64, device=dev)
    bufs = [ops.BatchBuffers(B, geom, M, dev) for _ in range(DEPTH)]
    gms = [torch.zeros((B, 4), dtype=torch.float64, device=dev) for _ in range(DEPTH)]
    mk = masks(mode)
    streams = [torch.cuda.Stream(device=dev) for _ in range(DEPTH)] if mk is None else [masked_stream(m) for m in mk]

    def step(i):
        k = i % DEPTH
        with torch.cuda.stream(streams[k]):
            ops.compress_batch(xyz, offs, tm, gms[k], bufs[k], ground_threshold=0.1, acc=0.04, ground_seed=0, frame_ids=fid)
    for i in range(8 * DEPTH):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ref = bufs[0].q16[:4].cpu().numpy().copy()
    return dt / steps * 1e3, ref


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    base = None
    print("| CU placement of the three pipeline slots | ms per step | outputs equal |")
    print("|---|---|---|")
    for rep in range(2):
        for mode in ("none", "two_thirds", "half", "disjoint", "xcd_mod", "xcd_div"):
            ms, ref = run(mode, steps)
            base = ref if base is None else base
            print("| %s | %.4f | %s |" % (mode, ms, bool(np.array_equal(ref, base))), flush=True)

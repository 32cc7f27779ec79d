"""Synthetic LiDAR sweeps for benchmarks and parity tests.

Not part of the reference: the reference's datalists point at files that do not exist here, so the
measured workload is a seeded synthetic scene of the same shape.  Scene: tilted ground plane, 40
axis-aligned boxes, 20 vertical cylinders, max range 80 m; one ray per lattice cell of the target
geometry with angular jitter, range noise and 12 % drop-outs; points shuffled, fp32 [N,3].

All randomness comes from NumPy PCG64(seed = 20000 + frame_id) on the host so a frame is the same on
every device; the ray casting itself runs in torch (fp64) on the requested device.
"""
import math

import numpy as np
import torch

MAX_RANGE = 80.0


def _scene(rng):
    roll, pitch = np.radians(rng.uniform(-1, 1, 2))
    n = np.array([math.sin(pitch), math.sin(roll), 1.0])
    n /= np.linalg.norm(n)
    d = -n[2] * (-1.73 + rng.normal(0, 0.02))
    nb, nc = 40, 20
    br, ba = rng.uniform(3, 60, nb), rng.uniform(0, 2 * np.pi, nb)
    bs = rng.uniform(0.5, 6, (nb, 3))
    bc = np.stack([br * np.cos(ba), br * np.sin(ba), -1.73 + bs[:, 2] / 2], 1)
    cr, ca = rng.uniform(3, 60, nc), rng.uniform(0, 2 * np.pi, nc)
    crad = rng.uniform(0.15, 1.0, nc)
    ch = rng.uniform(1, 8, nc)
    cyl = np.stack([cr * np.cos(ca), cr * np.sin(ca), crad, np.full(nc, -1.73), -1.73 + ch], 1)
    return n, d, bc - bs / 2, bc + bs / 2, cyl


SCENES = ("default", "shell", "noise", "corridor")


def make_frame(frame_id, H, W, vmax_deg=2.0, vmin_deg=-24.9, hfov_deg=360.0, device="cpu", scene="default"):
    """Return one synthetic sweep as a float32 torch tensor [N,3] on `device` (N ~ 0.8*H*W).
    scene: "default" (above), or one of the adversarial inputs of the FPS pruning study (the tile-pruned
    FPS is data dependent, the reference kernel is not -- ops/fps/src/sampling_gpu.cu:49-69):
      "shell"    every ray returns from a sphere of 30 m around the sensor (all candidates equally far: flat boxes, many ties in reach);
      "noise"    every pixel an independent range, uniform in 2 .. 80 m (every tile's box spans the whole radial extent: no pruning);
      "corridor" two walls 1.5 m left and right of the sensor over a floor, open ends (ranges from 1.5 m to the 80 m cut-off,
                 most pixels near the sensor)."""
    assert scene in SCENES, scene
    rng = np.random.Generator(np.random.PCG64(20_000 + int(frame_id)))
    n, d, bmin, bmax, cyl = _scene(rng)
    P = H * W
    jh = rng.uniform(-0.45, 0.45, P)
    jw = rng.uniform(-0.45, 0.45, P)
    noise = rng.normal(0, 0.015, P)
    keep = rng.random(P) >= 0.12
    perm = rng.permutation(P)

    dev = torch.device(device)
    t64 = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64, device=dev)
    hh = torch.arange(H, device=dev, dtype=torch.float64).repeat_interleave(W)
    ww = torch.arange(W, device=dev, dtype=torch.float64).repeat(H)
    vmin, vmax = math.radians(vmin_deg), math.radians(vmax_deg)
    el = vmin + (vmax - vmin) * (hh + t64(jh)) / (H - 1)
    az = math.radians(hfov_deg) * (ww + t64(jw)) / W
    dirs = torch.stack([torch.cos(el) * torch.cos(az), torch.cos(el) * torch.sin(az), torch.sin(el)], 1)  # [P,3]

    inf = torch.full((P,), float("inf"), dtype=torch.float64, device=dev)
    # ground
    den = dirs @ t64(n)
    tg = -d / den
    t = torch.where((tg > 0) & torch.isfinite(tg), tg, inf)
    # boxes (slab method, origin at sensor)
    inv = 1.0 / dirs  # [P,3]
    t0 = t64(bmin)[None] * inv[:, None, :]  # [P,nb,3]
    t1 = t64(bmax)[None] * inv[:, None, :]
    tn = torch.minimum(t0, t1).amax(-1)
    tf = torch.maximum(t0, t1).amin(-1)
    hit = (tf >= tn) & (tn > 0)
    tb = torch.where(hit, tn, torch.full_like(tn, float("inf"))).amin(-1)
    t = torch.minimum(t, tb)
    # vertical cylinders
    c = t64(cyl)
    a2 = (dirs[:, 0] ** 2 + dirs[:, 1] ** 2)[:, None]
    bq = -(dirs[:, 0:1] * c[None, :, 0] + dirs[:, 1:2] * c[None, :, 1])
    cq = (c[:, 0] ** 2 + c[:, 1] ** 2 - c[:, 2] ** 2)[None]
    disc = bq * bq - a2 * cq
    tc = (-bq - torch.sqrt(torch.clamp(disc, min=0))) / a2
    zc = tc * dirs[:, 2:3]
    ok = (disc > 0) & (tc > 0) & (zc >= c[None, :, 3]) & (zc <= c[None, :, 4])
    tc = torch.where(ok, tc, torch.full_like(tc, float("inf"))).amin(-1)
    t = torch.minimum(t, tc)

    if scene == "shell":
        t = torch.full((P,), 30.0, dtype=torch.float64, device=dev)
    elif scene == "noise":
        t = t64(rng.uniform(2.0, 80.0, P))
    elif scene == "corridor":
        tw = 1.5 / torch.abs(dirs[:, 1]).clamp(min=1e-12)                       # walls at y = -1.5 and y = +1.5
        tfl = torch.where(dirs[:, 2] < 0, -1.73 / dirs[:, 2].clamp(max=-1e-12), inf)    # floor at z = -1.73
        t = torch.minimum(tw, tfl)
    t = t + t64(noise)
    valid = torch.isfinite(t) & (t < MAX_RANGE) & (t > 0.5) & torch.as_tensor(keep, device=dev)
    xyz = (dirs * t[:, None]).to(torch.float32)
    order = torch.as_tensor(perm, device=dev)
    xyz = xyz[order][valid[order]]
    return xyz.contiguous()


def make_batch(frame_ids, H, W, device="cpu", **kw):
    """Concatenated batch: (xyz float32 [sum N, 3], offsets int64 [B+1]) on `device`."""
    frames = [make_frame(f, H, W, device=device, **kw) for f in frame_ids]
    offs = np.zeros(len(frames) + 1, np.int64)
    offs[1:] = np.cumsum([f.shape[0] for f in frames])
    return torch.cat(frames, 0), torch.as_tensor(offs, device=device)

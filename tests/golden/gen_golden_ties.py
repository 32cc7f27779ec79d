#!/usr/bin/env python3
"""Third fixture (round 5): the first-maximum tie order of the assignment, pinned by the GENUINE reference (build container only; needs
/root/reference and oracle/_ref).  The reference's own PointCloudSegment.segment(cpu=True) -- calc_plane_residual_depth,
calc_cluster_residual_radius, np.argmax(-np.abs(distance)), the relabel (utils/segment_utils.py:21-23,64-67,127-131,168-169) -- runs on a
CONSTRUCTED range image and a PRESCRIBED centre list (the FPS op is replaced by a function returning these indices; the ground model is
injected as in gen_golden.py):

  * rows of constant range, so the lattice puts pixels (almost) equally far from two centres of the same row: squared distances that
    differ in their last bits and round to ONE fp32 radius (np.linalg.norm) -- the argmax keeps the LOWER index, which is often not the
    one with the smaller squared distance;
  * pixels whose RANGE is moved onto the bisector of their two nearest centres and then walked ulp by ulp until the two squared distances
    differ but round to one radius, the lower index off the minimum (tune_ranges);
  * two centres listed twice (exact duplicates: the lower index wins everywhere);
  * empty pixels (label 1 after the relabel), whatever their distances say.

Stored: the inputs needed to rebuild the image (row ranges, empty-pixel stride, the tuned pixels and their ranges), the ground model, the 100 centre indices, and the labels the
reference produced (uint8).  Nothing of the reference travels."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  (stubs, write_lidar_yaml)

import numpy as np  # noqa: E402
import torch  # noqa: E402

orc = gg.orc
GEOM = "VelodyneVLP16"
EMPTY_STRIDE = 97


def build_image():
    """-> (row ranges f32 [H], ri f32 [H,W]); a pure function of constants (the tests rebuild it from the stored row ranges and stride)."""
    g = orc.GEOMS[GEOM]
    H, W = g["H"], g["W"]
    row_r = (9.0 + 2.5 * np.arange(H)).astype(np.float32)
    ri = np.repeat(row_r[:, None], W, 1).copy()
    ri.reshape(-1)[::EMPTY_STRIDE] = 0
    return row_r, ri


def choose_centres(ri, mask):
    """100 centre pixels among the reference's candidates (mask), spread at random, two of them listed twice."""
    rng = np.random.default_rng(515)
    ok = np.flatnonzero(mask & (ri.reshape(-1) != 0))
    idx = rng.choice(ok, 100, replace=False).astype(np.int64)
    idx[49] = idx[7]                            # exact duplicate of a lower-indexed centre
    idx[50] = idx[91]                           # ... and a lower-indexed copy of a higher one
    return idx.astype(np.int32)


def d2_rows(xyz, cen):
    d = xyz[:, None, :] - cen[None]
    return ((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]).astype(np.float32)


def tune_ranges(ri, tm, cen_pix, want=160):
    """Moves the RANGE of chosen pixels onto the bisector of their two nearest centres (a pixel's range touches no other pixel's distances, and
    the centres are pixels that stay as they are), then walks a few ulps around it until the two squared distances DIFFER but round to one fp32
    radius with the LOWER index off the minimum.  -> {pixel: range}"""
    rng = np.random.default_rng(77)
    flat, rays = ri.reshape(-1).copy(), tm.reshape(-1, 3)
    cen = (flat[cen_pix, None] * rays[cen_pix]).astype(np.float32)
    out = {}
    cset = set(int(c) for c in cen_pix)
    for p in rng.permutation(flat.size):
        if len(out) >= want:
            break
        if flat[p] == 0 or int(p) in cset:
            continue
        t = rays[p].astype(np.float64)
        d2 = d2_rows((flat[p] * rays[p]).astype(np.float32)[None], cen)[0]
        a, b = np.argsort(d2, kind="stable")[:2]
        if (cen[a] == cen[b]).all():
            continue
        ca, cb = cen[a].astype(np.float64), cen[b].astype(np.float64)
        den = 2.0 * (t @ (ca - cb))
        if abs(den) < 1e-6:
            continue
        r0 = np.float32(((ca @ ca) - (cb @ cb)) / den)
        if not (4.0 < r0 < 90.0):
            continue
        r = r0
        for _ in range(4):
            r = np.nextafter(r, np.float32(0))
        for _ in range(9):
            xyz = (r * rays[p]).astype(np.float32)[None]
            dd = d2_rows(xyz, cen)[0]
            k1 = int(dd.argmin())
            tie = (np.sqrt(dd) == np.sqrt(dd[k1])) & (dd != dd[k1])
            if tie[:k1].any():
                out[int(p)] = r
                break
            r = np.nextafter(r, np.float32(100))
    return out


def main():
    gg.install_stubs()
    from dataset.transformer import PCTransformer
    from utils.segment_utils import PointCloudSegment
    import ops.fps.fps_utils as fu
    g = orc.GEOMS[GEOM]
    tmp = "/tmp/rpcc_golden_tmp"
    os.makedirs(tmp, exist_ok=True)
    yml = os.path.join(tmp, GEOM + ".yaml")
    gg.write_lidar_yaml(yml, g)
    T = PCTransformer(yml, None)
    row_r, ri = build_image()
    ri3 = ri[..., None]
    pc = T.range_image_to_point_cloud(ri3)
    gm = np.array([0.01, -0.02, -0.9997, -1.72])
    ps = PointCloudSegment(T.transform_map)
    depth_dif = ps.calc_plane_residual_vertical(pc, gm)
    mask = (depth_dif > 0.1).reshape(-1)
    cen_pix = choose_centres(ri, mask)
    over = tune_ranges(ri, T.transform_map, cen_pix)
    for p_, r_ in over.items():
        ri.reshape(-1)[p_] = r_
    ri3 = ri[..., None]
    pc = T.range_image_to_point_cloud(ri3)
    depth_dif = ps.calc_plane_residual_vertical(pc, gm)
    mask = (depth_dif > 0.1).reshape(-1)
    assert mask[cen_pix].all(), "every prescribed centre must be a candidate of the reference (pc_left)"
    compact = np.cumsum(mask) - 1               # pixel -> index in pc_left (row-major compaction, utils/segment_utils.py:120)
    want = compact[cen_pix].astype(np.int32)
    fu.furthest_point_sample = lambda xyz, npoint: torch.from_numpy(want)[None]
    PointCloudSegment.ransac_plane_segmentation = staticmethod(lambda *a, **k: (None, gm.copy()))
    seg, gm_out = ps.segment(pc, ri3, {"segment_method": "FPS", "ground_vertical_threshold": 0.1, "cluster_num": 100, "DBSCAN_eps": 1.5},
                             cpu=True)
    assert np.array_equal(gm_out, gm)
    # how many pixels the fixture decides by tie order: another centre's fp32 radius equals the minimum's with a different squared distance
    cen = pc.reshape(-1, 3)[cen_pix]
    d = pc.reshape(-1, 1, 3) - cen[None]
    d2 = ((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]).astype(np.float32)
    rad = np.sqrt(d2)
    k1 = d2.argmin(1)
    m1 = d2[np.arange(d2.shape[0]), k1]
    tie = (rad == np.sqrt(m1)[:, None]) & (d2 != m1[:, None])
    lower = (tie & (np.arange(100)[None] < k1[:, None])).any(1) & (ri.reshape(-1) != 0)
    lab = seg.reshape(-1)
    decided = int((lower & (lab >= 2) & (lab - 2 != k1)).sum())
    print("pixels with a sqrt-level tie:", int(tie.any(1).sum()), "- with a LOWER index inside the tie:", int(lower.sum()),
          "- labelled with that lower index (not the arg-min of the squared distance):", decided)
    assert decided >= 20
    np.savez_compressed(os.path.join(HERE, "pins_ties_vlp16.npz"), row_ranges=row_r, empty_stride=np.int32(EMPTY_STRIDE), ground_model=gm,
                        centre_pixels=cen_pix, seg_idx=seg.astype(np.uint8),
                        tuned_pixels=np.array(sorted(over), np.int32), tuned_ranges=np.array([over[k] for k in sorted(over)], np.float32))
    json.dump(dict(geom=GEOM, sqrt_tie_pixels=int(tie.any(1).sum()), lower_index_in_tie=int(lower.sum()), decided_by_tie_order=decided,
                   labels=int(seg.max()) + 1, sha_seg_idx=gg.sha(seg.astype(np.uint8)), numpy=np.__version__),
              open(os.path.join(HERE, "pins_ties_manifest.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()

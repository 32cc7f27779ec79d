"""PointCloudSegment -- mirror of the reference's utils/segment_utils.py on the HIP path.

Same method names, argument meaning and return conventions (numpy in / numpy out), so
tools/compress.py-style drivers read the same.  What differs, by construction:

* `segment(..., cpu=...)`: both values run the kernels that reproduce the reference's cpu=True branch
  (utils/segment_utils.py:118-131); the reference's torch branch (:133-148) is numerically different
  and is not reproduced.
* the ground plane: the reference calls Open3D's random RANSAC; here `ransac_plane_segmentation` is the
  build's seeded RANSAC on the device (rpcc_ground_ransac).  Assigning a different callable to
  `PointCloudSegment.ransac_plane_segmentation` (as the reference's users can) injects a model.
* DBSCAN segmentation (utils/segment_utils.py:149-164) is out of scope.
"""
import numpy as np
import torch

from . import ops


class PointCloudSegment:
    def __init__(self, transform_map, plane_num=1, device="cuda:0", seed=0, frame_id=0):
        self.plane_num = plane_num
        self.transform_map = transform_map
        self.device = torch.device(device)
        self.seed = int(seed)
        self.frame_id = int(frame_id)   # identity of the frame for the seeded RANSACs (utils.frame_identity(path))
        self._tm = torch.from_numpy(np.ascontiguousarray(transform_map, dtype=np.float32)).to(self.device)
        self._cache = {}

    # -- host <-> device helpers ----------------------------------------------------------------
    def _ri(self, range_image):
        h, w = self.transform_map.shape[:2]
        return torch.from_numpy(np.ascontiguousarray(range_image, dtype=np.float32).reshape(1, h, w)).to(self.device)

    def _seg(self, seg_idx):
        """labels on the device: a byte up to cluster_num 254, uint16 above (the *_wide stage entries, up to 1022)"""
        a = np.ascontiguousarray(seg_idx)
        wide = int(a.max()) > 255 if a.size else False
        return torch.from_numpy(a.astype(np.uint16 if wide else np.uint8)[None]).to(self.device)

    # -- reference interface --------------------------------------------------------------------
    ransac_plane_segmentation = None  # assign a callable(points, threshold, ransac_n, num_iterations) to inject

    def segment(self, point_cloud, range_image, segment_cfg, cpu=True):
        """utils/segment_utils.py:95-170 -> (seg_idx int64 [H,W], ground_model f64 [4])."""
        assert self.transform_map is not None, "Must set transform_map first."
        method = segment_cfg["segment_method"]
        assert method in ["FPS", "DBSCAN"]
        if method != "FPS":
            raise NotImplementedError("DBSCAN segmentation is out of scope (SURVEY.md section 2)")
        thr = segment_cfg["ground_vertical_threshold"]
        # the stage entries: byte labels up to 254, uint16 ones (rpcc_assign_wide) up to 1022 -- a larger value is named in the error (BatchCompressor: 65533)
        M = ops.check_cluster_num(segment_cfg["cluster_num"], stage="mid")
        ri = self._ri(range_image)
        inject = type(self).ransac_plane_segmentation
        if inject is not None:
            # the reference's candidate selection (utils/segment_utils.py:101-106) feeding the injected fit
            pc_filter = point_cloud[np.where(point_cloud[..., 2] < -1.5)]
            if pc_filter.shape[0] > 5000:
                pc_filter = pc_filter[np.random.choice(pc_filter.shape[0], 5000, replace=False)]
            if pc_filter.shape[0] < 800:
                pc_filter = point_cloud.reshape((-1, 3))
            _, gm = inject(pc_filter)
            ground = torch.from_numpy(np.asarray(gm, np.float64).reshape(1, 4)).to(self.device)
        else:
            ground, _ = ops.ground_ransac(ri, self._tm, seed=self.seed, frame_ids=[self.frame_id])
        # RPCC_FPS_FMA / RPCC_FPS_TIE_CUDA (environment): the CUDA binary's contraction / tree tie rule (ops.fps_range)
        from ._lib import fps_mode_flags
        if fps_mode_flags(None, None):
            temp, info = ops.ground_mask(ri, self._tm, ground, thr, fps_table=False)
            cen_pix, centers = ops.fps_range(ri, self._tm, temp, info, M, fma=None, cuda_tie=None)
        else:
            temp, info, tab = ops.ground_mask(ri, self._tm, ground, thr, fps_table=True)
            cen_pix, centers = ops.fps_range(ri, self._tm, temp, info, M, fps_table=tab)
        seg = ops.assign(ri, self._tm, ground, centers)
        self._cache = dict(ri=ri, ground=ground, seg=seg, M=M)
        return seg[0].cpu().numpy().astype(np.int64), ground[0].cpu().numpy()

    def cluster_modeling(self, point_cloud, range_image, seg_idx, model_cfg, ground_model=None):
        """utils/segment_utils.py:172-217 -> float64 [max(seg), 4] (row k-1 models label k)."""
        method = model_cfg["model_method"]
        assert method in ["point", "plane"]
        ri, seg = self._ri(range_image), self._seg(seg_idx)
        M = max(int(seg_idx.max()) - 1, 1)
        nrow = int(seg_idx.max()) + 1
        ground = torch.zeros((1, 4), dtype=torch.float64, device=self.device)
        if method == "point":
            model, _ = ops.point_model(ri, seg, ground, M)
        else:
            model = ops.plane_model(ri, self._tm, seg, M, angle_threshold=model_cfg["angle_threshold"], seed=self.seed,
                                    frame_ids=[self.frame_id])
        return model[0, 1:nrow].cpu().numpy().astype(np.float64)

    def intra_predict(self, seg_idx, model_param):
        """utils/segment_utils.py:219-233 -> f32 [H,W,1]."""
        h, w = self.transform_map.shape[:2]
        K = max(int(model_param.shape[0]), 3)
        mp = np.zeros((1, K, 4), np.float32)
        mp[0, :model_param.shape[0]] = np.asarray(model_param).astype(np.float32)   # the pybind11 fp64->fp32 cast
        pred = ops.intra_predict(self._seg(seg_idx), torch.from_numpy(mp).to(self.device), self._tm)
        return pred[0].cpu().numpy().reshape(h, w, 1)

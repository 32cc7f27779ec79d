"""PCTransformer -- mirror of the reference's dataset/transformer.py on the HIP path.

Same constructor, attributes and method names; numpy in, numpy out like the pybind11 op it replaces.
The CSV beam-table variant (dataset/transformer.py:13-22,68-91) is not implemented: no registry entry of
the reference uses it (dataset/__init__.py:29-49).
"""
import numpy as np
import torch

from . import ops
from .utils import load_yaml


class PCTransformer:
    def __init__(self, lidar_cfg=None, channel_distribute_csv=None, device="cuda:0"):
        if channel_distribute_csv is not None:
            raise NotImplementedError("per-beam CSV tables (dataset/transformer.py:13-22) are out of scope")
        self.even_dist = True
        cfg = load_yaml(lidar_cfg) if isinstance(lidar_cfg, str) else dict(lidar_cfg)
        self.horizontal_FOV = cfg["HORIZONTAL_FOV"] * (np.pi / 180)      # dataset/transformer.py:32-37
        self.vertical_max = cfg["VERTICAL_ANGLE_MAX"] * (np.pi / 180)
        self.vertical_min = cfg["VERTICAL_ANGLE_MIN"] * (np.pi / 180)
        self.vertical_FOV = self.vertical_max - self.vertical_min
        self.H = int(cfg["RANGE_IMAGE_HEIGHT"])
        self.W = int(cfg["RANGE_IMAGE_WIDTH"])
        self.device = torch.device(device)
        self.transform_map = self.create_transform_map()
        self.geom = ops.make_geom(self.H, self.W, self.horizontal_FOV, self.vertical_max, self.vertical_min)
        self._tm_dev = None

    def create_transform_map(self):
        """dataset/transformer.py:41-54."""
        return ops.transform_map(self.H, self.W, self.horizontal_FOV, self.vertical_max, self.vertical_min)

    @property
    def tm_dev(self):
        if self._tm_dev is None:
            self._tm_dev = torch.from_numpy(self.transform_map).to(self.device)
        return self._tm_dev

    def point_cloud_to_range_image(self, point_cloud):
        """dataset/transformer.py:62-66: [N,3] -> f32 [H,W]."""
        xyz = torch.from_numpy(np.ascontiguousarray(point_cloud[:, :3], dtype=np.float32)).to(self.device)
        offs = torch.tensor([0, xyz.shape[0]], dtype=torch.int64, device=self.device)
        return ops.project(xyz, offs, self.geom)[0].cpu().numpy()

    def range_image_to_point_cloud(self, range_image):
        """dataset/transformer.py:94-101: [H,W] or [H,W,1] -> [H,W,3]."""
        if range_image.ndim not in (2, 3):
            assert False
        ri = torch.from_numpy(np.ascontiguousarray(range_image, dtype=np.float32).reshape(1, self.H, self.W)).to(self.device)
        return ops.backproject(ri, self.tm_dev)[0].cpu().numpy()

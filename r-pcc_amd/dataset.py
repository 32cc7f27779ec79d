"""Data access -- mirror of the reference's dataset/__init__.py:52-69 (build_dataset) and
dataset/dataset.py (DatasetTemplate): .bin / .npy / .txt readers, the lidar registry and the
range-image loader.  Dataset-specific converters (dataset/datasets/*.py) and the Open3D formats
(.ply/.pcd read, .pcd write) are out of scope."""
import os
import struct

import numpy as np

from .transformer import PCTransformer

_CFG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lidar_cfg")

__lidar_cfg__ = {
    "VelodyneVLP16": os.path.join(_CFG, "Velodyne_VLP_16.yaml"),
    "Velodyne32E": os.path.join(_CFG, "Velodyne_HDL_32E.yaml"),
    "Velodyne64E": os.path.join(_CFG, "Velodyne_HDL_64E.yaml"),
    "Velodyne64E_2048": os.path.join(_CFG, "Velodyne_HDL_64E_2048.yaml"),   # BASELINE synthetic geometry
}
__dataset_cfg__ = {
    "KITTI": __lidar_cfg__["Velodyne64E"],
    "KITTI_test": os.path.join(_CFG, "Velodyne_HDL_64E_unofficial.yaml"),
    "NCLT": __lidar_cfg__["Velodyne32E"],
    "Oxford": __lidar_cfg__["Velodyne32E"],
    "HKUSTCampus": __lidar_cfg__["VelodyneVLP16"],
}


class DatasetTemplate:
    def __init__(self, datalist, dataset_cfg, channel_distribute_csv=None, use_radius_outlier_removal=False,
                 device="cuda:0"):
        self.data_list = []
        if datalist is not None:
            with open(datalist, "r") as f:
                self.data_list = [line.strip() for line in f if line.strip()]
        if use_radius_outlier_removal:
            raise NotImplementedError("radius outlier removal needs Open3D (out of scope)")
        if dataset_cfg is not None:
            self.dataset_cfg = dataset_cfg
            self.PCTransformer = PCTransformer(dataset_cfg, channel_distribute_csv, device=device)
            self.transform_map = self.PCTransformer.transform_map

    def __len__(self):
        return len(self.data_list)

    def __getitem__(self, index):
        file_name = self.data_list[index]
        point_cloud, range_image, original = self.load_range_image_points_from_file(file_name)
        return point_cloud, range_image, original, file_name

    @staticmethod
    def load_data(file):
        """dataset/dataset.py:43-63 -> [N,3]."""
        ext = file.split(".")[-1]
        if ext == "txt":
            pc = np.loadtxt(file)
        elif ext == "bin":
            pc = np.fromfile(file, dtype=np.float32).reshape((-1, 4))
        elif ext in ("npy", "npz"):
            pc = np.load(file)
        else:
            assert False, "File type not correct: " + file
        return pc[:, :3]

    def load_range_image_points_from_file(self, file):
        """dataset/dataset.py:65-70."""
        original = self.load_data(file)
        range_image = np.expand_dims(self.PCTransformer.point_cloud_to_range_image(original), -1)
        point_cloud = self.PCTransformer.range_image_to_point_cloud(range_image)
        return point_cloud, range_image, original

    @staticmethod
    def save_point_cloud_to_file(file, point_cloud, color=None):
        """dataset/dataset.py:72-107 (txt / bin / npy / ply)."""
        ext = file.split(".")[-1]
        point_cloud = point_cloud[np.where(np.sum(point_cloud, -1) != 0)]
        if ext in ("txt", "bin", "npy", "npz"):
            pc4 = np.concatenate((point_cloud, np.zeros((point_cloud.shape[0], 1))), -1)
            if ext == "txt":
                np.savetxt(file, pc4)
            elif ext == "bin":
                pc4.astype(np.float32).tofile(file)
            else:
                np.save(file, pc4)
        elif ext == "ply":
            with open(file, "wb") as fid:
                fid.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\n"
                           "property float y\nproperty float z\nend_header\n" % point_cloud.shape[0]).encode())
                fid.write(np.ascontiguousarray(point_cloud[:, :3], dtype="<f4").tobytes())
        else:
            assert False, "File type not correct."


def build_dataset(datalist=None, dataset_name=None, lidar_type=None, use_radius_outlier_removal=False, device="cuda:0"):
    """dataset/__init__.py:52-69."""
    if dataset_name is not None:
        return DatasetTemplate(datalist, __dataset_cfg__[dataset_name], None, use_radius_outlier_removal, device)
    if lidar_type is not None:
        return DatasetTemplate(datalist, __lidar_cfg__[lidar_type], None, use_radius_outlier_removal, device)
    return DatasetTemplate(datalist, dataset_cfg=None, use_radius_outlier_removal=use_radius_outlier_removal)

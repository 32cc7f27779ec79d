"""GPU: one frame through the reference-style per-stage classes (the flow of tools/compress.py: numpy in, numpy out, one C-ABI call per stage),
warm, 20 repetitions -- where a single-frame caller's time goes.  Usage: python tools_dev/mirror_times.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rpcc_amd  # noqa: E402,F401
from rpcc_amd.compress_utils import BasicCompressor, QuantizationModule, compress_point_cloud  # noqa: E402
from rpcc_amd.dataset import build_dataset  # noqa: E402
from rpcc_amd.segment_utils import PointCloudSegment  # noqa: E402
from rpcc_amd.utils import load_compressor_cfg  # noqa: E402
import torch  # noqa: E402

z = np.load(os.path.join(ROOT, "tests", "golden", "example_64E.npz"))
xyz = np.ascontiguousarray(z["xyz"], np.float32)
dataset = build_dataset(lidar_type="Velodyne64E")
cfg = load_compressor_cfg(os.path.join(ROOT, "r-pcc_amd", "cfgs", "compressor.yaml")) if os.path.exists(os.path.join(ROOT, "r-pcc_amd", "cfgs", "compressor.yaml")) else None
segment_cfg = dict(segment_method="FPS", cluster_num=100, ground_vertical_threshold=0.1, DBSCAN_eps=0.5)
model_cfg = dict(model_method="point", angle_threshold=75)
pc_seg = PointCloudSegment(dataset.transform_map, seed=0, frame_id=7)
T = dataset.PCTransformer
acc = {}


def lap(name, t0):
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    acc.setdefault(name, []).append(t1 - t0)
    return t1


for rep in range(25):
    t = time.perf_counter()
    ri = np.expand_dims(T.point_cloud_to_range_image(xyz), -1); t = lap("point_cloud_to_range_image", t)   # (dataset.load_range_image_points_from_file hands [H,W,1])
    pc = T.range_image_to_point_cloud(ri); t = lap("range_image_to_point_cloud", t)
    seg_idx, ground_model = pc_seg.segment(pc, ri, segment_cfg, cpu=False); t = lap("segment", t)
    cm = pc_seg.cluster_modeling(pc, ri, seg_idx, model_cfg); t = lap("cluster_modeling", t)
    model_param = np.concatenate((ground_model.reshape(1, 4), cm), 0)
    pred = pc_seg.intra_predict(seg_idx, model_param); t = lap("intra_predict", t)
    residual = ri - pred; t = lap("residual (numpy)", t)
    QM = QuantizationModule(0.02, uniform=True)
    rq, sal, kp = QM.quantize_residual(residual, seg_idx, pc, ri); t = lap("quantize_residual", t)
for k, v in acc.items():
    print("%-32s %8.3f ms (first call %8.1f ms)" % (k, 1e3 * float(np.mean(v[5:])), 1e3 * v[0]))
print("sum %.3f ms per frame" % (1e3 * sum(float(np.mean(v[5:])) for v in acc.values())))

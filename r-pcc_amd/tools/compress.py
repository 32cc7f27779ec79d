#!/usr/bin/env python3
"""Single-frame compression -- same flags and stage structure as the reference's tools/compress.py,
running on the HIP path.  (--eval prints the depth-error check only: the chamfer / PSNR metrics of the
reference need packages that are out of scope.)"""
import argparse
import os
import sys
import time

BASE_DIR = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, BASE_DIR)

import numpy as np  # noqa: E402

import rpcc_amd  # noqa: E402,F401
from rpcc_amd.compress_utils import (BasicCompressor, QuantizationModule, compress_point_cloud,  # noqa: E402
                                     decompress_point_cloud, read_compressed_bitstream, save_compressed_bitstream)
from rpcc_amd.dataset import build_dataset  # noqa: E402
from rpcc_amd.segment_utils import PointCloudSegment  # noqa: E402
from rpcc_amd.utils import frame_identity, load_compressor_cfg  # noqa: E402

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_parser(datalist=False):
    p = argparse.ArgumentParser()
    if datalist:
        p.add_argument("--datalist", help="datalist of point cloud files.")
        p.add_argument("--output_dir", help="output folder.")
        p.add_argument("--workers", type=int, default=1, help="host threads for entropy coding and file output (tools/compress_datalist.py:25).")
        p.add_argument("--output", action="store_true", help="print per-frame information.")
        p.add_argument("--batch", type=int, default=64, help="frames per device batch.")
        p.add_argument("--points-per-frame", dest="points_per_frame", type=int, default=None,
                       help="initial staging capacity in points per frame (default H x W; the staging slots grow on demand).")
        p.add_argument("--ingest", choices=("auto", "rows", "xyz"), default="auto",
                       help="rows: .bin sweeps go to the device as stored (x, y, z, intensity rows, 16-byte stride, no host pass); "
                            "xyz: loaded and sliced on the host (12 bytes per point over the link); auto: rows when every file is a .bin.")
        p.add_argument("--gather", action="store_true",
                       help="multi-GPU: every rank entropy-codes its shard, the .rpcc bytes are gathered to rank 0 (RCCL) in "
                            "datalist order and rank 0 writes all files (default: every rank writes its own files).")
        p.add_argument("--gather-round", dest="gather_round", type=int, default=4096, help="--gather: frames per rank and round.")
    else:
        p.add_argument("--input", help="single frame input for static compression.")
        p.add_argument("--output", help="output bitstream.")
    p.add_argument("--lidar", help="lidar type of this point cloud collection.")
    p.add_argument("--compressor_yaml", default=os.path.join(PKG, "cfgs/compressor.yaml"))
    p.add_argument("--basic_compressor", type=str, default=None, help="for manual setting.")
    p.add_argument("--accuracy", type=float, default=None, help="for manual setting.")
    p.add_argument("--segment_method", type=str, default=None, help="for manual setting.")
    p.add_argument("--cluster_num", type=int, default=None, help="for manual setting.")
    p.add_argument("--DBSCAN_eps", type=float, default=None, help="for manual setting.")
    p.add_argument("--model_method", type=str, default=None, help="for manual setting.")
    p.add_argument("--angle_threshold", type=float, default=None, help="for manual setting.")
    p.add_argument("--nonuniform", action="store_true", help="for manual setting.")
    p.add_argument("--eval", action="store_true", help="evaluate the reconstruction quality.")
    p.add_argument("--cpu", action="store_true", help="accepted for compatibility; the HIP path has no CPU mode.")
    p.add_argument("--seed", type=int, default=0, help="seed of the ground / plane RANSAC (this build).")
    p.add_argument("--fps_fma", type=int, default=None, choices=(0, 1, 2),
                   help="FPS distance as the reference's CUDA binary may contract it (0 un-fused = default, 1 fma(dz,dz,fma(dx,dx,dy*dy)), "
                        "2 fma(dz,dz,fma(dy,dy,dx*dx))); sets RPCC_FPS_FMA.")
    p.add_argument("--fps_tie_cuda", action="store_true",
                   help="FPS ties between exactly equal distances resolved like the CUDA kernel's reduction tree (default: lowest "
                        "index); sets RPCC_FPS_TIE_CUDA.")
    return p


def apply_fps_mode(args):
    """--fps_fma / --fps_tie_cuda -> the environment variables every FPS call of this build reads (ops.compress_batch,
    segment_utils.PointCloudSegment.segment)."""
    if getattr(args, "fps_fma", None) is not None:
        os.environ["RPCC_FPS_FMA"] = str(args.fps_fma)
    if getattr(args, "fps_tie_cuda", False):
        os.environ["RPCC_FPS_TIE_CUDA"] = "1"


def resolve_cfg(args):
    """tools/compress.py:45-84: YAML values overridden by the individual flags."""
    cfg = load_compressor_cfg(args.compressor_yaml)
    accuracy = cfg["accuracy"] * 2
    segment_cfg = {"segment_method": cfg["segment_method"], "ground_vertical_threshold": cfg["ground_threshold"],
                   "cluster_num": cfg["cluster_num"], "DBSCAN_eps": cfg["DBSCAN_eps"]}
    model_cfg = {"model_method": cfg["modeling_method"], "angle_threshold": cfg["plane_angle_threshold"]}
    bc = BasicCompressor(compressor_yaml=args.compressor_yaml)
    if args.basic_compressor is not None:
        bc.set_method(args.basic_compressor)
    if args.accuracy is not None:
        accuracy = args.accuracy * 2
    for key, dst, name in (("segment_method", segment_cfg, "segment_method"), ("cluster_num", segment_cfg, "cluster_num"),
                           ("DBSCAN_eps", segment_cfg, "DBSCAN_eps"), ("model_method", model_cfg, "model_method"),
                           ("angle_threshold", model_cfg, "angle_threshold")):
        if getattr(args, key) is not None:
            dst[name] = getattr(args, key)
    uniform = False if args.nonuniform else cfg["compress_framework"] == "uniform"
    return cfg, accuracy, segment_cfg, model_cfg, bc, uniform


def make_quantizer(cfg, accuracy, uniform):
    if uniform:
        return QuantizationModule(accuracy)
    return QuantizationModule(accuracy, uniform=False, level_kp_num=tuple(cfg["level_key_point_num"]),
                              level_dacc=tuple(cfg["level_delta_acc"]), ground_salience_level=cfg["ground_salience_level"],
                              feature_region=cfg["feature_region"], segments=cfg["segments"], sharp_num=cfg["sharp_num"],
                              less_sharp_num=cfg["less_sharp_num"], flat_num=cfg["flat_num"])


def compress_wide(args, cfg, accuracy, segment_cfg, model_cfg, basic_compressor, uniform):
    """cluster_num above 254 (labels as uint16): the per-stage mirror classes keep labels in a byte, so the frame goes through the batch front-end
    (pipeline.BatchCompressor -> rpcc_compress_batch_wide) as a batch of one.  Same container, same decoder."""
    from rpcc_amd.pipeline import BatchCompressor
    dataset = build_dataset(lidar_type=args.lidar)
    t0 = time.time()
    frame = dataset.load_data(args.input)
    bc = BatchCompressor(dataset.PCTransformer, cluster_num=segment_cfg["cluster_num"], accuracy=accuracy / 2,
                         ground_threshold=segment_cfg["ground_vertical_threshold"], uniform=uniform, model_method=model_cfg["model_method"],
                         compressor_cfg=dict(cfg), basic_compressor=basic_compressor.method_name, seed=args.seed)
    blob = bc.compress([frame], frame_ids=[frame_identity(args.input)])[0]
    with open(args.output, "wb") as f:
        f.write(blob)
    t1 = time.time()
    buf = bc._buf
    point_num = int((buf.ri[0] != 0).sum().item())
    print("\nCompression finished (cluster_num = %d: uint16 labels, batch front-end)." % segment_cfg["cluster_num"])
    print("binary bitstream save in ", args.output)
    print("    Total time cost: ", t1 - t0)
    bits = os.path.getsize(args.output) * 8
    print("\nCompression Results: ")
    print("    Compression ratio: ", (point_num * 32 * 3) / bits)
    print("    BPP: ", bits / point_num)
    if args.eval:
        from rpcc_amd.tools.decompress import decode_frame
        level_acc = np.array([accuracy] * len(cfg["level_key_point_num"])) + np.array(cfg["level_delta_acc"])
        rec, _, _ = decode_frame(read_compressed_bitstream(args.output, uniform=uniform), basic_compressor, dataset.PCTransformer,
                                 segment_cfg["cluster_num"], accuracy, level_acc, uniform, want_points=False)
        ri = buf.ri[0].cpu().numpy()
        dif = np.abs(rec - ri)   # every pixel, empty ones included, as compress() below and tools/compress.py:172-181
        bound = accuracy + (0.0 if uniform else 0.06) + 0.00001
        print("\nReconstruction quality: ")
        print("    Depth Error (mean): ", float(np.mean(dif)))
        print("    Depth Error (max): ", float(np.max(dif)))
        if float(np.max(dif)) > bound:
            raise AssertionError("Reconstruction error... Please check...")


def compress(args):
    apply_fps_mode(args)
    cfg, accuracy, segment_cfg, model_cfg, basic_compressor, uniform = resolve_cfg(args)
    from rpcc_amd import ops
    if ops.is_wide(ops.check_cluster_num(segment_cfg["cluster_num"])):
        return compress_wide(args, cfg, accuracy, segment_cfg, model_cfg, basic_compressor, uniform)
    dataset = build_dataset(lidar_type=args.lidar)
    model_num = segment_cfg["cluster_num"] + 1
    pc_seg = PointCloudSegment(dataset.transform_map, seed=args.seed, frame_id=frame_identity(args.input))

    t_init = time.time()
    point_cloud, range_image, original_point_cloud = dataset.load_range_image_points_from_file(args.input)
    point_num = int((point_cloud[..., 0] != 0).sum())
    t_load_data = time.time()
    seg_idx, ground_model = pc_seg.segment(point_cloud, range_image, segment_cfg, cpu=args.cpu)
    t_segmentation = time.time()
    cluster_models = pc_seg.cluster_modeling(point_cloud, range_image, seg_idx, model_cfg)
    model_param = np.concatenate((ground_model.reshape(1, 4), cluster_models), 0)
    t_modeling = time.time()
    range_image_pred = pc_seg.intra_predict(seg_idx, model_param)
    residual = range_image - range_image_pred
    t_intra_pred = time.time()
    QM = make_quantizer(cfg, accuracy, uniform)
    residual_quantized, salience_level, key_point_map = QM.quantize_residual(residual, seg_idx, point_cloud, range_image)
    t_quantization = time.time()
    original_data, compressed_data = compress_point_cloud(basic_compressor, model_param, seg_idx, salience_level,
                                                          residual_quantized, full=False)
    t_basic_compressor = time.time()
    save_compressed_bitstream(args.output, compressed_data, uniform=uniform)
    t_save = time.time()

    print("\nCompression finished.")
    print("binary bitstream save in ", args.output)
    print("\nTime Cost:")
    print("    Load data: ", t_load_data - t_init)
    print("    Segmentation module: ", t_segmentation - t_load_data)
    print("    Modeling module: ", t_modeling - t_segmentation)
    print("    Intra-prediction module: ", t_intra_pred - t_modeling)
    print("    Quantization module: ", t_quantization - t_intra_pred)
    print("    Basic compressor module (", basic_compressor.method_name, "): ", t_basic_compressor - t_quantization)
    print("    Save binary file: ", t_save - t_basic_compressor)
    print("    Total time cost: ", t_save - t_init)
    print("    Total time cost without loading data: ", t_save - t_load_data)
    bits = os.path.getsize(args.output) * 8
    print("\nCompression Results: ")
    print("    Compression ratio: ", (point_num * 32 * 3) / bits)
    print("    BPP: ", bits / point_num)
    print("\n")

    if args.eval:
        cd = read_compressed_bitstream(args.output, uniform=uniform)
        rq, seg2, sal2, plane_param = decompress_point_cloud(cd, basic_compressor, model_param.shape[0],
                                                             dataset.transform_map.shape[0], dataset.transform_map.shape[1])
        QM2 = make_quantizer(cfg, accuracy, uniform)
        res2 = QM2.dequantize_residual(rq, seg2, sal2)
        rec = pc_seg.intra_predict(seg2, plane_param) + res2
        dif = np.abs(rec - range_image)
        bound = accuracy + (0.0 if uniform else 0.06) + 0.00001
        print("\nReconstruction quality: ")
        print("    Depth Error (mean): ", float(np.mean(dif)))
        print("    Depth Error (max): ", float(np.max(dif)))
        if float(np.max(dif)) > bound:
            raise AssertionError("Reconstruction error... Please check...")


if __name__ == "__main__":
    a = make_parser().parse_args()
    print("Input arguments:")
    for key, val in vars(a).items():
        print("{:16} {}".format(key, val))
    compress(a)

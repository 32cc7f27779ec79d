#!/usr/bin/env python3
"""Datalist decompression -- counterpart of the reference's tools/decompress_datalist.py: every
`.rpcc` named in the datalist is decoded and written as <output_dir>/<path>.bin."""
import os
import sys
from concurrent import futures

BASE_DIR = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, BASE_DIR)

import numpy as np  # noqa: E402

import rpcc_amd  # noqa: E402,F401
from rpcc_amd.compress_utils import read_compressed_bitstream  # noqa: E402
from rpcc_amd.dataset import build_dataset  # noqa: E402
from rpcc_amd.sharding import shard_indices  # noqa: E402
from rpcc_amd.tools.compress import make_parser, resolve_cfg  # noqa: E402
from rpcc_amd.tools.decompress import decode_frame  # noqa: E402


def decompress(args):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    device = "cuda:%d" % int(os.environ.get("LOCAL_RANK", "0"))
    cfg, accuracy, segment_cfg, model_cfg, basic_compressor, uniform = resolve_cfg(args)
    dataset = build_dataset(datalist=args.datalist, lidar_type=args.lidar, device=device)
    level_acc = np.array([accuracy] * len(cfg["level_key_point_num"])) + np.array(cfg["level_delta_acc"])
    for i in shard_indices(len(dataset), rank, world):
        name = dataset.data_list[i]
        cd = read_compressed_bitstream(name, uniform=uniform)
        rec, pc, _ = decode_frame(cd, basic_compressor, dataset.PCTransformer, segment_cfg["cluster_num"], accuracy,
                                  level_acc, uniform)
        rel = name[1:] if name.startswith("/") else name
        out = os.path.join(args.output_dir, rel)
        out = out.replace(out.split(".")[-1], "bin")
        os.makedirs(os.path.dirname(out), exist_ok=True)
        dataset.save_point_cloud_to_file(out, pc.reshape(-1, 3))
        if args.output:
            print("%s -> %s (%d points)" % (name, out, int((rec != 0).sum())))


if __name__ == "__main__":
    a = make_parser(datalist=True).parse_args()
    decompress(a)

#!/usr/bin/env python3
"""Single-frame decompression -- counterpart of the reference's tools/decompress.py on the HIP path.
Nothing about the configuration is stored in the .rpcc file: the decoder needs the same YAML / flags /
lidar type as the encoder (as in the reference)."""
import os
import sys
import time

BASE_DIR = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, BASE_DIR)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import rpcc_amd  # noqa: E402,F401
from rpcc_amd import ops  # noqa: E402
from rpcc_amd.compress_utils import decompress_point_cloud, read_compressed_bitstream  # noqa: E402
from rpcc_amd.dataset import build_dataset  # noqa: E402
from rpcc_amd.tools.compress import make_parser, resolve_cfg  # noqa: E402


def decode_frame(blob_dict, basic_compressor, transformer, cluster_num, accuracy, level_acc, uniform, want_points=True):
    """decompress_point_cloud + dequantise + predict + back-project (tools/decompress.py:79-112)."""
    H, W = transformer.H, transformer.W
    d = basic_compressor.decompress_dict(blob_dict)
    # The .rpcc file stores no configuration (as in the reference): a wrong --lidar / cluster_num / framework shows up as
    # payload sizes that do not fit.  Check them here instead of letting the kernels index past their buffers.
    P, K = H * W, cluster_num + 2
    if len(d["plane_param"]) % 16 != 0:
        raise ValueError("plane_param payload is not a whole number of float32 [.,4] rows")
    plane_param = np.frombuffer(d["plane_param"], dtype=np.float32).reshape(-1, 4)
    if plane_param.shape[0] > K:
        raise ValueError("bitstream holds %d model rows, the configuration allows cluster_num + 2 = %d" % (plane_param.shape[0], K))
    if len(d["contour_map"]) != (P + 7) // 8:
        raise ValueError("contour_map holds %d bytes, a %dx%d range image needs %d (wrong --lidar?)"
                         % (len(d["contour_map"]), H, W, (P + 7) // 8))
    if len(d["idx_sequence"]) % 2 or len(d["residual_quantized"]) % 2:
        raise ValueError("idx_sequence / residual_quantized payloads are not 16-bit arrays")
    s_chk = np.frombuffer(d["idx_sequence"], dtype=np.uint16)
    n_contour = int(np.unpackbits(np.frombuffer(d["contour_map"], dtype=np.uint8))[:P].sum())
    if s_chk.size != n_contour:
        raise ValueError("idx_sequence holds %d labels, the contour map marks %d runs" % (s_chk.size, n_contour))
    if s_chk.size and int(s_chk.max()) >= plane_param.shape[0]:
        raise ValueError("idx_sequence names label %d but only %d model rows are stored" % (int(s_chk.max()), plane_param.shape[0]))
    if not uniform:
        if "salience_level" not in d:
            raise ValueError("non-uniform framework: the bitstream holds no salience_level payload (written with the uniform framework?)")
        sl_chk = np.frombuffer(d["salience_level"], dtype=np.uint8)
        if sl_chk.size > K:
            raise ValueError("salience_level holds more entries than labels")
        if sl_chk.size < plane_param.shape[0]:
            raise ValueError("salience_level holds %d entries for %d model rows" % (sl_chk.size, plane_param.shape[0]))
        if sl_chk.size and int(sl_chk.max()) >= len(level_acc):
            raise ValueError("salience level %d in the bitstream, the configuration defines %d levels (other level_key_point_num?)"
                             % (int(sl_chk.max()), len(level_acc)))
    dev = transformer.device
    model = torch.zeros((1, K, 4), dtype=torch.float32, device=dev)
    model[0, : plane_param.shape[0]] = torch.from_numpy(plane_param.copy()).to(dev)
    bits = torch.from_numpy(np.frombuffer(d["contour_map"], dtype=np.uint8).copy()[None]).to(dev)
    seq = torch.zeros((1, H * W), dtype=torch.uint16, device=dev)
    s = np.frombuffer(d["idx_sequence"], dtype=np.uint16)
    seq[0, : s.size] = torch.from_numpy(s.copy()).to(dev)
    seg = ops.contour_decode(bits, seq, H, W, cluster_num)
    q = torch.zeros((1, H * W), dtype=torch.int16, device=dev)
    qq = np.frombuffer(d["residual_quantized"], dtype=np.int16)
    n_nonempty = int(((seg.view(torch.int16) if seg.dtype == torch.uint16 else seg) != 1).sum().item())   # (uint16 labels: cluster_num > 254)
    if qq.size != n_nonempty:
        raise ValueError("residual_quantized holds %d values, the label map has %d non-empty pixels" % (qq.size, n_nonempty))
    q[0, : qq.size] = torch.from_numpy(qq.copy()).to(dev)
    if uniform:
        rec, pc = ops.decode(seg, q, model, transformer.tm_dev, accuracy, want_points=want_points)
    else:
        sal = torch.zeros((1, K), dtype=torch.uint8, device=dev)
        sl = np.frombuffer(d["salience_level"], dtype=np.uint8)
        sal[0, : sl.size] = torch.from_numpy(sl.copy()).to(dev)
        rec, pc = ops.decode(seg, q, model, transformer.tm_dev, list(level_acc), salience=sal, want_points=want_points)
    return rec[0].cpu().numpy(), (pc[0].cpu().numpy() if pc is not None else None), seg[0].cpu().numpy()


def decompress(args):
    cfg, accuracy, segment_cfg, model_cfg, basic_compressor, uniform = resolve_cfg(args)
    dataset = build_dataset(lidar_type=args.lidar)
    level_acc = np.array([accuracy] * len(cfg["level_key_point_num"])) + np.array(cfg["level_delta_acc"])
    t0 = time.time()
    cd = read_compressed_bitstream(args.input, uniform=uniform)
    rec, pc, seg = decode_frame(cd, basic_compressor, dataset.PCTransformer, segment_cfg["cluster_num"], accuracy,
                                level_acc, uniform)
    t1 = time.time()
    dataset.save_point_cloud_to_file(args.output, pc.reshape(-1, 3))
    print("\nDecompression finished.")
    print("reconstructed point cloud save in ", args.output)
    print("    Decode time: ", t1 - t0)
    print("    Points: ", int((rec != 0).sum()))


if __name__ == "__main__":
    p = make_parser()
    a = p.parse_args()
    print("Input arguments:")
    for key, val in vars(a).items():
        print("{:16} {}".format(key, val))
    decompress(a)

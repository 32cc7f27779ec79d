"""ContourExtractor -- mirror of the reference's utils/contour_utils.py:178-230 (row-wise run
boundaries) on the HIP path.  The unused 2-D FloodFill variants (utils/contour_utils.py:8-175) are out of
scope."""
import numpy as np
import torch

from . import ops


class ContourExtractor:
    device = "cuda:0"

    @staticmethod
    def extract_contour(idx_map):
        """-> (contour_map int32 [H,W] of 0/1, idx_sequence int32 [n]) like contour_utils_cpp.extract_contour."""
        seg = torch.from_numpy(np.ascontiguousarray(idx_map).astype(np.uint8)[None]).to(ContourExtractor.device)
        bits, seq, nseq = ops.contour_encode(seg)
        h, w = idx_map.shape
        cm = np.unpackbits(bits[0].cpu().numpy())[: h * w].reshape(h, w).astype(np.int32)
        return cm, seq[0, : int(nseq[0])].cpu().numpy().astype(np.int32)

    @staticmethod
    def recover_map(contour_map, idx_sequence):
        """-> idx_map int32 [H,W] like contour_utils_cpp.recover_map."""
        h, w = contour_map.shape
        dev = ContourExtractor.device
        bits = torch.from_numpy(np.packbits(np.ascontiguousarray(contour_map).astype(bool), axis=None)[None]).to(dev)
        seq = torch.zeros((1, h * w), dtype=torch.uint16, device=dev)
        s = torch.from_numpy(np.ascontiguousarray(idx_sequence).astype(np.uint16))
        seq[0, : s.numel()] = s.to(dev)
        return ops.contour_decode(bits, seq, h, w)[0].cpu().numpy().astype(np.int32)

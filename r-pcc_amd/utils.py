"""Config loading (mirror of the reference's utils/utils.py:18-25)."""
import os
import zlib

import yaml


class Config(dict):
    """dict with attribute access (what the reference gets from EasyDict)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    __setattr__ = dict.__setitem__


def load_yaml(path):
    with open(path, "r") as f:
        return Config(yaml.safe_load(f))


def load_compressor_cfg(yaml_file):
    """utils/utils.py:18-25: YAML -> attribute dict."""
    return load_yaml(yaml_file)


def frame_identity(path):
    """Stable identity of a frame for the seeded RANSACs of this build (rpcc_ground_ransac / rpcc_plane_model:
    frame_ids): CRC-32 of the file's base name.  A file's planes -- and so its .rpcc bytes -- then do not depend on the
    batch size, the position in the batch, the number of ranks or the tool (compress.py / compress_datalist.py) that
    processes it.  (The reference's Open3D RANSAC is unseeded: its output is not reproducible at all.)"""
    return zlib.crc32(os.path.basename(str(path).strip()).encode()) & 0x7FFFFFFF

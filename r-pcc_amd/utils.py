"""Config loading (mirror of the reference's utils/utils.py:18-25)."""
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

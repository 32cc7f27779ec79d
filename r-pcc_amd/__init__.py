"""rpcc_amd -- MI355X-native implementation of the R-PCC per-frame compression hot path.

Directory name is `r-pcc_amd/`; import it as `rpcc_amd` (see /rpcc_amd.py at the repo root).
"""
__version__ = "0.1.0"

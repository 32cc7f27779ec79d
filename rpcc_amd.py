"""Import alias: the package directory is `r-pcc_amd/` (a hyphen is not importable), so
`import rpcc_amd` loads that directory as the package `rpcc_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "r-pcc_amd")
_spec = importlib.util.spec_from_file_location(
    "rpcc_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["rpcc_amd"] = _mod
_spec.loader.exec_module(_mod)

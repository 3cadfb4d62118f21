"""Import shim: loads the package that lives in `neural-tape-modeling_amd/` (a hyphen is not a valid
Python identifier) under the importable name `ntm_amd`, including its submodules."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "neural-tape-modeling_amd")
_spec = importlib.util.spec_from_file_location("ntm_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_pkg = importlib.util.module_from_spec(_spec)
sys.modules["ntm_amd"] = _pkg
_spec.loader.exec_module(_pkg)

"""Import shim: the package directory is named ``phantom-fhe_amd`` (hyphen, as the project layout
prescribes), which Python cannot import by name.  ``import phantom_fhe_amd`` loads it."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "phantom-fhe_amd")
_spec = importlib.util.spec_from_file_location(
    "phantom_fhe_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["phantom_fhe_amd"] = _mod
_spec.loader.exec_module(_mod)

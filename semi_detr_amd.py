"""Import shim: the package directory is ``semi-detr_amd/`` (not a valid Python identifier), so
``import semi_detr_amd`` loads it from there and installs it under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "semi-detr_amd")
_spec = importlib.util.spec_from_file_location(
    "semi_detr_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["semi_detr_amd"] = _mod
_spec.loader.exec_module(_mod)

"""Import shim: the package directory is ``nerf-pytorch_amd/`` (hyphen, not importable by
name); ``import nerf_pytorch_amd`` loads it from there and installs it in sys.modules."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nerf-pytorch_amd")
_spec = importlib.util.spec_from_file_location("nerf_pytorch_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["nerf_pytorch_amd"] = _mod
_spec.loader.exec_module(_mod)

"""Import shim: ``import adunet_amd`` loads the package that lives in the directory
``adaptive-depth-u-net-for-image-super-resolution-segmentation_amd/`` (not a valid Python
identifier because of the hyphens) under the importable name ``adunet_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                    "adaptive-depth-u-net-for-image-super-resolution-segmentation_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)

"""Import alias: the package directory is named ``voxel-raycaster_amd`` (hyphen),
which Python cannot import by name; ``import voxel_raycaster_amd`` loads it."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "voxel-raycaster_amd")
_spec = importlib.util.spec_from_file_location(
    "voxel_raycaster_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["voxel_raycaster_amd"] = _mod
_spec.loader.exec_module(_mod)

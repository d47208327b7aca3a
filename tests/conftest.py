import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def atlas():
    import scenes
    return scenes.hash_atlas()


@pytest.fixture(scope="session")
def gpu_available():
    import voxel_raycaster_amd as vrc
    import ctypes
    n = ctypes.c_int32()
    vrc.lib.vrc_device_count(ctypes.byref(n))
    return n.value > 0

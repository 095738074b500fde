import os
import sys

import pytest

try:  # one HIP runtime per process: torch's bundled runtime must be the first one loaded (see voxhip._preload_torch_hip_runtime)
    import torch  # noqa: F401
except ImportError:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "raytracing-voxilizer-vulkan-intresection_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Build (or reuse) libvoxhip.so and the oracle; both are compiled in-tree."""
    import __graft_entry__ as ge
    ge.build()
    return True


@pytest.fixture(scope="session")
def vx(built):
    import voxhip
    voxhip.lib()
    return voxhip


@pytest.fixture(scope="session")
def gpu(vx):
    if vx.device_count() < 1:
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box (libvoxhip has no CPU path)")
    return vx

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_visible():
    """True if HIP sees at least one device (checked through the runtime, without importing torch)."""
    import ctypes
    for name in ("libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
        try:
            hip = ctypes.CDLL(name)
            n = ctypes.c_int(0)
            return hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
        except OSError:
            continue
    return False


def pytest_collection_modifyitems(config, items):
    # `pytest tests` on a box without a GPU: the -m gpu tests are skipped (with the reason), not failed
    if any("gpu" in it.keywords for it in items) and not _gpu_visible():
        skip = pytest.mark.skip(reason="no HIP device visible: -m gpu tests need a real MI355X (the product has no CPU fallback)")
        for it in items:
            if "gpu" in it.keywords:
                it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def s4p_lib_built():
    from super4pcs_amd import build as B
    return B.build()

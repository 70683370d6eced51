import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    """True when a HIP device is visible (hipGetDeviceCount through the runtime the product library uses); no torch import needed."""
    import ctypes
    for name in ("libamdhip64.so", "libamdhip64.so.7", "/opt/rocm/lib/libamdhip64.so"):
        try:
            hip = ctypes.CDLL(name)
        except OSError:
            continue
        n = ctypes.c_int(0)
        try:
            return hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
        except Exception:
            return False
    return False


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests are the device parity tests: on a host without a GPU they are skipped, not failed (the driver's CPU tier runs
    `-m "not gpu"`; a plain `pytest tests/` on a CPU box used to report 40 hard failures)."""
    if not any("gpu" in item.keywords for item in items):
        return
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no ROCm-capable device on this host (gpu-marked test)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    import oracle_ctypes
    oracle_ctypes.build()
    return oracle_ctypes

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
    """True when this host exposes an AMD GPU to compute: the kernel driver's /dev/kfd node is there and usable.  Deliberately NOT a
    hipGetDeviceCount call: loading a HIP runtime here, before the product library has mapped the one it shares with torch
    (librabft_simulator_amd/_lib.py _one_hip_runtime), puts two ROCr runtimes into the process and the second finds no device."""
    return os.path.exists("/dev/kfd") and os.access("/dev/kfd", os.R_OK | os.W_OK)


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

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_visible() -> bool:
    """A HIP device is usable on this box: the kernel driver node exists and the runtime counts >= 1
    device.  torch.cuda.device_count() does not initialise the GPU on this image."""
    if not os.path.exists("/dev/kfd"):
        return False
    try:
        import torch
        return torch.cuda.device_count() > 0
    except Exception:  # noqa: BLE001
        return False


def pytest_collection_modifyitems(config, items):
    """Plain `pytest` on a box without a GPU skips the gpu-marked tests instead of failing them with
    BN_ERR_NO_DEVICE (the product has no CPU path to fall back to)."""
    if _gpu_visible():
        return
    skip = pytest.mark.skip(reason="no HIP device on this box (gpu-marked tests run with -m gpu on an MI355X)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.lib()  # builds liboracle.so on first use
    return oracle


@pytest.fixture(scope="session")
def bnlib():
    """The product library; built in-tree by __graft_entry__.build().  No fallback."""
    from bayesiannetwork_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.lib()

import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def golden(name: str):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def lib():
    """libfdmi.so, built in-tree if hipcc is available and the .so is stale/missing."""
    from foldingdiff_amd import _binding, build

    try:
        build.build()
    except RuntimeError as e:  # no hipcc on this machine: use the shipped .so if any
        if not os.path.exists(_binding.LIB_PATH):
            pytest.fail(f"libfdmi.so is not built and cannot be built here: {e}")
    return _binding.load()


@pytest.fixture(scope="session")
def gpu(lib):
    if lib.fd_device_count() < 1:
        pytest.fail("test marked gpu but no HIP device is visible")
    return 0

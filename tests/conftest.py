import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_ready():
    """A HIP device and the built library: what every `gpu` test needs."""
    try:
        import torch
        if not torch.cuda.is_available():
            return False
    except Exception:
        return False
    import atracdenc_amd
    return os.path.exists(atracdenc_amd.LIB_PATH)


def pytest_collection_modifyitems(config, items):
    # a plain `pytest tests/` on a box without a GPU skips the gpu-marked tests instead of failing them; an explicit
    # `-m gpu` run is never softened (there a missing device or library must be loud)
    if "gpu" in (config.getoption("-m") or ""):
        return
    if _gpu_ready():
        return
    skip = pytest.mark.skip(reason="needs an MI355X and the built libat3hip.so (run with -m gpu on the GPU box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    import at3_testlib
    at3_testlib.build_oracle()
    return at3_testlib.oracle()


@pytest.fixture(scope="session")
def golden_encode():
    import numpy as np
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "encode.npz"))


@pytest.fixture(scope="session")
def golden_stages():
    import numpy as np
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "stages.npz"))

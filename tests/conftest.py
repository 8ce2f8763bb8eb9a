import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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

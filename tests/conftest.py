import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with `pytest -m gpu` via gpurun)")


@pytest.fixture(scope="session")
def golden_pair():
    import numpy as np
    d = np.load(os.path.join(ROOT, "tests", "golden", "euroc_pair.npz"))
    return d["a"], d["b"]


@pytest.fixture(scope="session")
def klt_expected():
    import numpy as np
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "klt_expected.npz")))

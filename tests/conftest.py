import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with `pytest -m gpu` via gpurun)")
    # The synthetic sequences are rendered by forked workers where that is safe: on a host without a GPU (nothing in the process will
    # touch the HIP runtime -- forking after it has started is not).  Same pixels either way (harness/scene.py); the CPU suite is
    # ~10 minutes with one renderer, of which rendering is the larger part.
    if not os.path.exists("/dev/kfd"):
        os.environ.setdefault("XRSLAM_AMD_RENDER_WORKERS", str(max(1, min(8, len(os.sched_getaffinity(0))))))


def _pair(name):
    import numpy as np
    d = np.load(os.path.join(ROOT, "tests", "golden", name))
    return d["a"], d["b"]


@pytest.fixture(scope="session")
def golden_pair():
    """The reference's two test frames, undistorted by the restatement that reproduces its known answers exactly
    (tests/golden/make_golden.py, oracle/undistort.py)."""
    return _pair("euroc_pair.npz")


@pytest.fixture(scope="session")
def klt_expected():
    import numpy as np
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "klt_expected.npz")))


@pytest.fixture(scope="session")
def golden_pair_v1():
    """The same frames with the undistortion map rounded straight from the double (36 pixels differ): a second real
    image pair for the HIP-vs-oracle parity test."""
    return _pair("euroc_pair_v1.npz")


@pytest.fixture(scope="session")
def klt_expected_v1():
    import numpy as np
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "klt_expected_v1.npz")))

"""The CPU reference pipeline against committed trajectories (tests/golden/pipeline_poses.npz, made by
tests/golden/make_pipeline_poses.py): both threading modes, 64 frames of the synthetic S1 stream.  The pipeline's other tests
compare it with itself (GPU build vs CPU build, run vs run); this one notices when an edit of the host code changes what it
computes -- counters exactly, poses to 1e-9 (the arithmetic is deterministic; the slack is for a different libm)."""
import os

import numpy as np
import pytest

from tests.golden import make_pipeline_poses as gen

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pipeline_poses.npz")


@pytest.mark.parametrize("mode,name", [(0, "inline"), (1, "pipelined")])
def test_cpu_pipeline_reproduces_the_committed_trajectory(mode, name):
    if not os.path.exists(gen.ORACLE_LIB):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.join(gen.ROOT, "oracle")])
    g = np.load(GOLDEN)
    poses, counts = gen.run(mode)
    np.testing.assert_array_equal(counts, g[name + "_counts"])
    assert poses.shape == g[name + "_poses"].shape
    np.testing.assert_array_equal(poses[:, 0], g[name + "_poses"][:, 0])
    np.testing.assert_allclose(poses[:, 1:], g[name + "_poses"][:, 1:], rtol=1e-9, atol=1e-12)

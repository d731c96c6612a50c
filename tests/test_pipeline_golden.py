"""The CPU reference pipeline against committed trajectories (tests/golden/pipeline_poses.npz, made by
tests/golden/make_pipeline_poses.py): both threading modes, 64 frames of the synthetic S1 stream.  The pipeline's other tests
compare it with itself (GPU build vs CPU build, run vs run); this one notices when an edit of the host code changes what it
computes -- counters exactly, poses to 1e-9 (the arithmetic is deterministic; the slack is for a different libm)."""
import os

import numpy as np
import pytest

from tests.golden import make_pipeline_poses as gen

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pipeline_poses.npz")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


def test_getters_between_frames_do_not_move_the_pipelined_trajectory():
    """XRSLAMAmdFlush / XRSLAMGetResult(XRSLAM_RESULT_BIAS) wait for the backend job in flight; they must not PUBLISH its state
    (that happens at the next frame's hand-off, host/pipeline.hpp System::publish_backend_state), or the feature tracker of the
    next frame would start from the state of frame t instead of t-1 and the trajectory would depend on which getters a caller
    uses.  A run that calls both after every frame answers the committed pipelined trajectory."""
    import ctypes as C
    from xrslam_amd.harness import runner, scene
    if not os.path.exists(gen.ORACLE_LIB):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.join(gen.ROOT, "oracle")])
    g = np.load(GOLDEN)
    seq = scene.make_sequence(n_frames=gen.N_FRAMES, seed=1)
    s = runner.Session(gen.ORACLE_LIB, seq, threading=1)
    bias = (C.c_double * 6)()           # XRSLAMIMUBias: two XRSLAMBias of three doubles
    k = 0
    while s.step():
        assert not s.error(), s.error()
        s.sync()
        if k % 2:
            s.api.get_result(5, C.byref(bias))      # XRSLAM_RESULT_BIAS
        k += 1
    s.flush()
    s.sync()
    t = s.times()
    counts = np.array([t.frames, t.solves, t.solve_iterations, t.marginalizations, t.keyframes], np.int64)
    poses = np.array(s.poses)
    s.close()
    np.testing.assert_array_equal(counts, g["pipelined_counts"])
    np.testing.assert_allclose(poses[:, 1:], g["pipelined_poses"][:, 1:], rtol=1e-9, atol=1e-12)


def test_overlapped_detection_order_is_the_reference_order_bit_for_bit():
    """Inline tracking runs the detection's host half (and the rotation-misalignment test that sets FT_NO_TRANSLATION) beside
    localize_newframe's solve: the frame is mirrored first, the new keypoints and the tag reach both copies afterwards
    (pipeline.hpp, feature_tracker_work).  XRSLAM_AMD_NO_DETECT_OVERLAP=1 restores the reference's order -- detect, then mirror.
    Same poses, bit for bit, on a stream with a rotation-only phase (the tag is taken, subframes are merged and re-attached)."""
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from tests.test_swt_model import PausingTrajectory, ORACLE_LIB, SLAM
from xrslam_amd.harness import runner, scene
seq = scene.make_sequence(n_frames=150, seed=1, traj=PausingTrajectory(9.0, 10.6))
s = runner.Session(ORACLE_LIB, seq, slam_yaml=SLAM)
while s.step():
    assert not s.error(), s.error()
s.flush()
t = s.times()
np.save(sys.argv[1], np.array(s.poses))
print(t.frames, t.solves, t.keyframes, t.marginalizations)
s.close()
''' % ROOT
    outs = []
    for env_extra in ({}, {"XRSLAM_AMD_NO_DETECT_OVERLAP": "1"}):
        env = dict(os.environ)
        env.pop("XRSLAM_AMD_DUMP_SWT", None)
        env.update(env_extra)
        path = os.path.join(ROOT, "gpurun_out", "overlap_order_%d.npy" % len(outs))
        os.makedirs(os.path.dirname(path), exist_ok=True)
        r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append((np.load(path), r.stdout.strip().splitlines()[-1]))
    assert outs[0][1] == outs[1][1]                      # frames, solves, keyframes, marginalisations
    assert len(outs[0][0]) >= 100
    np.testing.assert_array_equal(outs[0][0], outs[1][0])

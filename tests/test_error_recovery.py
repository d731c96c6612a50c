"""What the C API's catch-all leaves behind (xrslam_api.cpp guarded() -> System::recover_after_error).

The reference aborts or throws through its C boundary; this library records the error and goes on, so an error in the middle of a
frame must not poison the frames that follow.  The CPU reference build (the product's host sources over oracle/xrhip_shim.cpp, which
keeps the product's begin / end state machine of the asynchronous entry points) lets a test make the n-th call of an inner entry
point fail like a device error would (orc_inject_failure) and then watch the stream:

  * a failure in the feature tracker loses that frame only;
  * a failure inside the sliding-window tracker (mirror_frame's integration, a solve) -- inline or on the backend thread -- takes the
    reference's tracking-failure branch (core/frontend_worker.cpp:75-81): back to the initialiser, tracking again a few frames later;
  * an error raised on the API thread while the backend job of the previous frame is running (an image with an unsupported channel
    count, XRSLAMManager.cpp:126-129) touches nothing of that job: the trajectory is bit-identical to a run without the bad calls.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from xrslam_amd.harness import runner, scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_LIB = os.path.join(ROOT, "oracle", "_build", "libxrslam_oracle.so")
BENCH_YAML = os.path.join(ROOT, "configs", "bench_slam_150.yaml")
N = 130
FAIL_PREINT_END, FAIL_SOLVE, FAIL_TRACK, FAIL_PREINT_BEGIN = 1, 2, 3, 4


@pytest.fixture(scope="module")
def seq():
    if not os.path.exists(ORACLE_LIB):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    return scene.make_sequence(n_frames=N, seed=3, workers=max(1, min(8, len(os.sched_getaffinity(0)))))


def _state(s):
    st = C.c_int(-1)
    s.api.get_result(runner.XRSLAM_RESULT_STATE, C.byref(st))
    return st.value


def _run(seq, mode, fail_at=None, which=0, countdown=1, bad_push_at=()):
    """-> (poses, per-frame tracking state, error text seen right after the failing frame, frames counted by the library)"""
    s = runner.Session(ORACLE_LIB, seq, slam_yaml=BENCH_YAML, threading=mode, init_frames=N)   # re-initialisation finds a state at any time
    s.lib.orc_inject_failure.argtypes = [C.c_int, C.c_int]
    s.lib.orc_inject_failure.restype = None
    states, err_after = [], ""
    k = 0
    while True:
        if k == fail_at:
            s.lib.orc_inject_failure(which, countdown)
        if k in bad_push_at:   # two-channel image: "Image channel is not supported!" on the API thread, the previous frame's job still running
            fr = seq["frames"][k]
            img = runner.XRSLAMImage(fr.ctypes.data, float(seq["cam_t"][k]), fr.strides[0], 0, 2, None)
            s.api.push(runner.XRSLAM_SENSOR_CAMERA, C.byref(img))
        if not s.step():
            break
        k += 1
        states.append(_state(s))
        if fail_at is not None and k in (fail_at + 1, fail_at + 2, fail_at + 3) and not err_after:
            err_after = s.error()
    s.flush()
    s.sync()
    frames = s.times().frames
    poses = np.array(s.poses)
    final_err = s.error()
    s.lib.orc_inject_failure(0, 0)
    s.close()
    return poses, states, err_after, frames, final_err


@pytest.fixture(scope="module")
def clean(seq):
    return {mode: _run(seq, mode) for mode in (0, 1)}


def _ate_after(poses, seq, t_from):
    late = [list(p) for p in poses if p[0] > t_from]
    return runner.ate_rmse(late, seq), len(late)


def test_clean_runs_track_throughout(seq, clean):
    for mode in (0, 1):
        poses, states, _, frames, err = clean[mode]
        assert err == "" and frames == N
        assert states[50:] == [1] * (N - 50)
        assert runner.ate_rmse(list(poses), seq) < 0.03


@pytest.mark.parametrize("mode", [0, 1], ids=["inline", "pipelined"])
def test_a_failed_lk_call_loses_one_frame(seq, clean, mode):
    poses, states, err, frames, _ = _run(seq, mode, fail_at=70, which=FAIL_TRACK)
    assert "injected device failure" in err and "xrhip_image_track" in err
    assert frames == N - 1                                   # the frame whose tracking failed never joined the tracking map
    assert states[60:] == [1] * (N - 60)                     # the sliding-window tracker was not involved: no re-initialisation
    ate, n = _ate_after(poses, seq, seq["cam_t"][75])
    assert n > 40 and ate < 0.03


@pytest.mark.parametrize("mode", [0, 1], ids=["inline", "pipelined"])
@pytest.mark.parametrize("which,countdown", [(FAIL_SOLVE, 1), (FAIL_SOLVE, 2), (FAIL_PREINT_END, 2), (FAIL_PREINT_END, 3),
                                             (FAIL_PREINT_BEGIN, 2)],
                         ids=["first-solve", "second-solve", "preint-end-2", "preint-end-3", "preint-begin-2"])
def test_a_failure_inside_the_sliding_window_tracker_reinitialises(seq, clean, mode, which, countdown):
    """Whatever the interrupted call was -- localize_newframe's solve, refine_subwindow's, the collection of mirror_frame's queued
    integration -- the stream is tracking again within a dozen frames and accurate afterwards; before the change the prepared / pending
    bookkeeping outlived the cancelled batches and every later frame failed with "nothing in flight"."""
    poses, states, err, frames, final_err = _run(seq, mode, fail_at=70, which=which, countdown=countdown)
    assert "injected device failure" in err
    assert final_err == err                                  # ONE error: nothing after it was refused
    assert frames >= N - 1
    assert states[-30:] == [1] * 30                          # tracking again (through the initialiser if the backend was hit)
    ate, n = _ate_after(poses, seq, seq["cam_t"][95])
    assert n >= 30 and ate < 0.03


@pytest.mark.parametrize("mode", [0, 1], ids=["inline", "pipelined"])
def test_an_api_thread_error_does_not_touch_the_frame_in_flight(seq, clean, mode):
    """Unsupported images pushed between frames (pipelined: while the backend job of the previous frame runs): the error is reported,
    the trajectory is the clean run's bit for bit."""
    poses, states, _, frames, err = _run(seq, mode, bad_push_at=(55, 56, 80, 81, 82, 100))
    assert err == "Image channel is not supported!"
    assert frames == N
    np.testing.assert_array_equal(poses, clean[mode][0])

"""GPU library vs CPU reference pipeline ON THE STREAMS bench.py TIMES.

bench.py's S1 line times frames 106..256 (default) or 56..76 (the driver's --steps 20 --warmup 5) of: the synthetic S1 stream,
seed 1, configs/bench_slam_150.yaml, initial states for the first 60 frames, host images through XRSLAM_SENSOR_CAMERA.  The
other pipeline tests stop at 64-100 frames of euroc_slam.yaml; here the same stream, configuration and call sequence run for 320
frames (>= 50 marginalisations) in BOTH threading modes, and the S2 line's stream (stress_slam_300.yaml, the fast trajectory,
seed 1) for 200 frames: identical discrete counters (frames, solves, dogleg iterations, marginalisations, keyframes) and poses
within north_star's 1e-4 relative of the CPU pipeline over the oracle (oracle/_build/libxrslam_oracle.so) in the same mode.
The CPU half alone (no GPU) pins the stream's shape: the window is full and marginalising once per keyframe over the frames the
bench times.

Round 4: north_star's WHOLE output list is compared, frame by frame, not only body poses and five
counters -- both builds write XRSLAM_AMD_DUMP_OUT (tests/outlog.py): per tracked frame the key points' pixel positions and the
track ids they carry, per backend frame the newest frame's pose / velocity / gyroscope and accelerometer bias, the window's
keyframe and subframe ids, the window-map track of every key point, and every track's id, tags, inverse depth and landmark --
ids / indices / tags EQUAL, floats within 1e-4 relative; XRSLAM_RESULT_BIAS is read after every frame and XRSLAM_RESULT_LANDMARKS at
the checkpoints through the C API (XRSLAMManager.cpp:153-236).  And the S3 line's stream (large_slam_600.yaml, 1280x720, 600
features, 20-keyframe window) runs for 150 frames: bench.py --workload s3 times frames >= 96, with the window full and
marginalising in-stream."""
import ctypes as C
import json
import os
import tempfile

import numpy as np
import pytest

from tests import outlog
from xrslam_amd.harness import runner, scene
from xrslam_amd.harness.trajectory import Trajectory

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_LIB = os.path.join(ROOT, "oracle", "_build", "libxrslam_oracle.so")
BENCH_YAML = os.path.join(ROOT, "configs", "bench_slam_150.yaml")
STRESS_YAML = os.path.join(ROOT, "configs", "stress_slam_300.yaml")
LARGE_YAML = os.path.join(ROOT, "configs", "large_slam_600.yaml")
LARGE_SENSOR = os.path.join(ROOT, "configs", "large_sensor_1280.yaml")
N_S1, N_S2, N_S3 = 320, 200, 150
XRSLAM_RESULT_LANDMARKS, XRSLAM_RESULT_BIAS = 3, 5
WORKERS = max(1, min(8, len(os.sched_getaffinity(0))))


@pytest.fixture(scope="module", autouse=True)
def _oracle_built():
    if not os.path.exists(ORACLE_LIB):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


class _Landmarks(C.Structure):
    _fields_ = [("landmarks", C.POINTER(C.c_double)), ("num_landmarks", C.c_int)]


def _run(lib_path, seq, yaml, mode, checkpoints=(), sensor_yaml=None):
    """The harness's step loop (the six reference symbols, host image); counters sampled after the frames in `checkpoints`.
    -> (poses, counters, counters at the checkpoints, output log, XRSLAM_RESULT_BIAS after every frame, XRSLAM_RESULT_LANDMARKS
    at the checkpoints)."""
    fd, out_path = tempfile.mkstemp(prefix="xr_out_", suffix=".bin")
    os.close(fd)
    os.environ["XRSLAM_AMD_DUMP_OUT"] = out_path          # read when the session's pipeline is constructed
    try:
        kw = {"sensor_yaml": sensor_yaml} if sensor_yaml else {}
        s = runner.Session(lib_path, seq, slam_yaml=yaml, threading=mode, **kw)
    finally:
        del os.environ["XRSLAM_AMD_DUMP_OUT"]
    marks, bias, landmarks = {}, [], {}
    k = 0
    while s.step():
        assert not s.error(), s.error()
        k += 1
        b = (C.c_double * 6)()                            # XRSLAMIMUBias: acc_bias, gyr_bias (joins the backend job, moves nothing)
        s.api.get_result(XRSLAM_RESULT_BIAS, C.byref(b))
        bias.append(list(b))
        if k in checkpoints:
            s.sync()
            t = s.times()
            marks[k] = (t.frames, t.solves, t.solve_iterations, t.marginalizations, t.keyframes)
            lm = _Landmarks()
            s.api.get_result(XRSLAM_RESULT_LANDMARKS, C.byref(lm))
            landmarks[k] = np.ctypeslib.as_array(lm.landmarks, (lm.num_landmarks, 3)).copy() if lm.num_landmarks else np.zeros((0, 3))
    s.flush()
    s.sync()
    assert not s.error(), s.error()
    t = s.times()
    counts = (t.frames, t.solves, t.solve_iterations, t.marginalizations, t.keyframes)
    poses = np.array(s.poses)
    s.close()
    log = outlog.read(out_path)
    os.unlink(out_path)
    return poses, counts, marks, log, np.array(bias), landmarks


def _assert_same_run(got, want, seq, ate_bound, name):
    poses_h, counts_h, marks_h, log_h, bias_h, lm_h = got
    poses_o, counts_o, marks_o, log_o, bias_o, lm_o = want
    if counts_h != counts_o or poses_h.shape != poses_o.shape or not np.allclose(poses_h, poses_o, rtol=1e-4, atol=1e-6):
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        np.savez_compressed(os.path.join(ROOT, "gpurun_out", "bench_stream_mismatch_%d.npz" % len(poses_o)), ph=poses_h, po=poses_o,
                            ch=np.array(counts_h), co=np.array(counts_o))
    assert marks_h == marks_o                      # the discrete decisions agree at every checkpoint, not only in total
    assert counts_h == counts_o
    assert poses_h.shape == poses_o.shape
    np.testing.assert_array_equal(poses_h[:, 0], poses_o[:, 0])
    np.testing.assert_allclose(poses_h[:, 1:], poses_o[:, 1:], rtol=1e-4, atol=1e-6)
    assert runner.ate_rmse(list(poses_h), seq) < ate_bound
    # north_star's output list, frame by frame: ids / indices equal, floats within 1e-4 relative (tests/outlog.py)
    stats = outlog.compare(log_h, log_o)
    assert stats["frames"] == counts_o[0] and stats["backend_frames"] > 0.8 * (counts_o[0] - 40)
    assert stats["keypoints"] > 100 * stats["backend_frames"] and stats["landmarks"] > 100 * stats["backend_frames"]
    # the C API's getters: biases after every frame, landmarks at the checkpoints
    assert bias_h.shape == bias_o.shape
    np.testing.assert_allclose(bias_h, bias_o, rtol=1e-4, atol=1e-6)
    assert sorted(lm_h) == sorted(lm_o)
    for k in lm_o:
        assert lm_h[k].shape == lm_o[k].shape and len(lm_o[k]) > 100, "XRSLAM_RESULT_LANDMARKS after frame %d" % k
        np.testing.assert_allclose(lm_h[k], lm_o[k], rtol=1e-4, atol=1e-4)
    stats["name"] = name
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "stream_parity_%s.json" % name), "w") as fh:
        json.dump(stats, fh, indent=1)


def test_output_log_round_trip_and_the_cpu_pipeline_is_reproducible(s1_seq_short):
    """Two CPU runs of the same stream write byte-identical output logs (the comparison below is between two builds, not two moods),
    and the log holds what it claims: one 'F' record per frame, one 'B' per frame behind the seeded window, ids that only grow."""
    a = _run(ORACLE_LIB, s1_seq_short, BENCH_YAML, 0)
    b = _run(ORACLE_LIB, s1_seq_short, BENCH_YAML, 0)
    st = outlog.compare(a[3], b[3], rtol=0, atol_px=0, atol_state=0, atol_point=0)
    assert st["keypoints_bit_identical"] == st["keypoints"] and st["landmarks_bit_identical"] == st["landmarks"]
    F, B = a[3]
    assert len(F) == a[1][0] == 70 and [f["id"] for f in F] == list(range(1, 71))
    assert len(B) >= 30 and all(x["id"] < y["id"] for x, y in zip(B, B[1:]))
    last = B[-1]
    assert len(last["window"]) == 10 or last["window"]                # keyframe ids ascending, subframes behind their keyframe
    ids = [w for w, _ in last["window"]]
    assert ids == sorted(ids) and all(all(sf > w for sf in subs) for w, subs in last["window"])
    carried = F[-1]["track"][F[-1]["track"] >= 0]
    assert len(carried) > 50 and len(set(carried)) == len(carried)   # a track id sits on one key point of a frame
    # XRSLAM_RESULT_BIAS after the k-th camera push == (ba, bg) of frame k - 1 (a frame is processed when the first later IMU sample arrives)
    prev = [b for b in B if b["id"] == 69][0]
    np.testing.assert_array_equal(a[4][-1], np.r_[prev["state"][13:16], prev["state"][10:13]])


# ------------------------------------------------------------------------------------------------ S1 (BASELINE config 2)
S1_MARKS = (56, 76, 106, 206, 256)      # the ends of the windows bench.py times (driver: 56..76; default: 106..256)


@pytest.fixture(scope="module")
def s1_seq():
    return scene.make_sequence(n_frames=N_S1, seed=1, workers=WORKERS)


@pytest.fixture(scope="module")
def s1_seq_short(s1_seq):
    return {k: (v[:70] if k in ("frames", "cam_t", "states") else v) for k, v in s1_seq.items()}


@pytest.fixture(scope="module")
def s1_cpu(s1_seq):
    return {mode: _run(ORACLE_LIB, s1_seq, BENCH_YAML, mode, S1_MARKS) for mode in (0, 1)}


def test_cpu_pipeline_is_in_steady_state_over_the_timed_frames(s1_seq, s1_cpu):
    """By frame 56 (4 * window + 16) the window is full and the first marginalisation is behind; from there on one
    marginalisation per keyframe -- >= 50 of them by the end of the run; the trajectory stays accurate throughout."""
    for mode in (0, 1):
        poses, counts, marks = s1_cpu[mode][:3]
        frames, solves, iters, margs, kfs = counts
        assert frames == N_S1 and margs >= 50
        f0, f1 = marks[56], marks[256]
        assert f0[3] >= 1                                  # the eigen-path one-off is inside the pre-roll
        assert f1[3] - f0[3] == f1[4] - f0[4] >= 40        # steady state: every keyframe pushes one out
        assert runner.ate_rmse(list(poses), s1_seq) < 0.03


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1], ids=["inline", "pipelined"])
def test_gpu_matches_cpu_reference_on_the_s1_bench_stream(s1_seq, s1_cpu, mode):
    from xrslam_amd import _lib
    _assert_same_run(_run(_lib.LIB_PATH, s1_seq, BENCH_YAML, mode, S1_MARKS), s1_cpu[mode], s1_seq, 0.03,
                     "s1_%s" % ("pipelined" if mode else "inline"))


# ------------------------------------------------------------------------------------------------ S2 (BASELINE config 3)
S2_MARKS = (76, 126, 176)


@pytest.fixture(scope="module")
def s2_seq():
    return scene.make_sequence(n_frames=N_S2, seed=1, traj=Trajectory(amp=1.5, speed=1.0, rot=0.8), workers=WORKERS)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1], ids=["inline", "pipelined"])
def test_gpu_matches_cpu_reference_on_the_s2_bench_stream(s2_seq, mode):
    """Round 5: the pipelined mode -- whose `variants` the S2 line prints -- under the same output-log parity."""
    from xrslam_amd import _lib
    want = _run(ORACLE_LIB, s2_seq, STRESS_YAML, mode, S2_MARKS)
    assert want[1][3] >= 20 and want[1][0] == N_S2         # marginalising: the 15-keyframe window is full by frame 76
    _assert_same_run(_run(_lib.LIB_PATH, s2_seq, STRESS_YAML, mode, S2_MARKS), want, s2_seq, 0.06,
                     "s2_%s" % ("pipelined" if mode else "inline"))


# ------------------------------------------------------------------------------------------------ S3 (BASELINE config 5)
S3_MARKS = (96, 116, 146)               # bench.py --workload s3 never times a frame before 4 * 20 + 16 = 96


@pytest.fixture(scope="module")
def s3_seq():
    return scene.make_sequence(n_frames=N_S3, seed=1, w=1280, h=720, K=(780.0, 778.0, 640.0, 360.0), workers=WORKERS)


class _Lazy(dict):
    """CPU reference runs by threading mode, computed on first use (the CPU-only suite needs the inline run only)."""

    def __init__(self, make):
        super().__init__()
        self._make = make

    def __missing__(self, mode):
        self[mode] = self._make(mode)
        return self[mode]


@pytest.fixture(scope="module")
def s3_cpu_by_mode(s3_seq):
    return _Lazy(lambda mode: _run(ORACLE_LIB, s3_seq, LARGE_YAML, mode, S3_MARKS, sensor_yaml=LARGE_SENSOR))


@pytest.fixture(scope="module")
def s3_cpu(s3_cpu_by_mode):
    return s3_cpu_by_mode[0]


def test_cpu_pipeline_s3_stream_has_a_full_window_over_the_timed_frames(s3_seq, s3_cpu):
    """The frames bench.py --workload s3 times (96 on) run with the 20-keyframe window full and one marginalisation per keyframe."""
    poses, counts, marks, log, _, _ = s3_cpu
    frames, solves, iters, margs, kfs = counts
    assert frames == N_S3
    f0, f1 = marks[96], marks[146]
    assert f0[3] >= 1                                       # the first marginalisation (eigen path) is inside the pre-roll
    assert f1[3] - f0[3] == f1[4] - f0[4] >= 8              # >= 8 in-stream marginalisations inside the checked steady state
    B = log[1]
    assert all(len(b["window"]) == 20 for b in B if b["id"] > 96)
    assert runner.ate_rmse(list(poses), s3_seq) < 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1], ids=["inline", "pipelined"])
def test_gpu_matches_cpu_reference_on_the_s3_bench_stream(s3_seq, s3_cpu_by_mode, mode):
    from xrslam_amd import _lib
    got = _run(_lib.LIB_PATH, s3_seq, LARGE_YAML, mode, S3_MARKS, sensor_yaml=LARGE_SENSOR)
    _assert_same_run(got, s3_cpu_by_mode[mode], s3_seq, 0.05, "s3_%s" % ("pipelined" if mode else "inline"))

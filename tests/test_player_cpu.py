"""BASELINE config 1 -- "CPU reference via the player (plumbing, no GPU)": the headless player's own sources linked
against the CPU reference library (oracle/_build/xrslam-player-ref, built by oracle/Makefile) on a synthetic ASL
directory.  Exercises, without a device, everything between the data set on disk and the trajectory file: the
reference player's command line, CSV / PNG readers, event order, undistortion, the XRSLAM.h call sequence, the TUM and
CSV writers and the ATE -- and pins the player to the ctypes harness running the same CPU library."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLAYER_REF = os.path.join(ROOT, "oracle", "_build", "xrslam-player-ref")
REF_LIB = os.path.join(ROOT, "oracle", "_build", "libxrslam_oracle.so")
SLAM = os.path.join(ROOT, "configs", "bench_slam_150.yaml")
SENSOR = os.path.join(ROOT, "configs", "euroc_sensor.yaml")


@pytest.fixture(scope="module")
def player():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    return PLAYER_REF


def _run(cmd):
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])


def test_cpu_reference_player_on_a_synthetic_euroc_directory(player, tmp_path):
    from xrslam_amd.harness import euroc, runner, scene
    seq = scene.make_sequence(n_frames=70, seed=5)
    root = euroc.write_euroc(seq, str(tmp_path / "mav0"))
    tum, csv = str(tmp_path / "traj.tum"), str(tmp_path / "traj.csv")
    res = _run([player, "-sc", SLAM, "-dc", SENSOR, "--tum", tum, "--csv", csv, "-p", "--no-undistort",
                "--bootstrap-frames", "60", "euroc://" + root])
    assert res["error"] == "" and res["frames"] == 70 and res["bootstrap_states"] == 60 and res["init_attempts"] == 0
    assert res["tracked"] >= 25                       # the first 36 frames seed the window
    assert 0 <= res["ate_rmse_m"] < 0.03
    rows = np.loadtxt(tum)
    assert rows.shape == (res["tracked"], 8) and np.all(np.diff(rows[:, 0]) > 0)
    np.testing.assert_allclose(np.linalg.norm(rows[:, 4:8], axis=1), 1.0, atol=1e-6)
    np.testing.assert_array_equal(np.loadtxt(csv, delimiter=","), rows)
    # the same CPU library driven through ctypes (in-memory frames and IMU samples): the same trajectory up to the
    # nanosecond text round trip of the time stamps
    sess = runner.Session(REF_LIB, seq, slam_yaml=SLAM, sensor_yaml=SENSOR)
    for _ in range(70):
        sess.step()
    ref = np.array([ps for ps in sess.poses if abs(ps[4]) + abs(ps[5]) + abs(ps[6]) + abs(ps[7]) > 0])
    sess.close()
    n = min(len(ref), len(rows))
    assert n >= 25
    assert np.abs(ref[-n:, 1:4] - rows[-n:, 1:4]).max() < 5e-3


def test_cpu_reference_player_pipelined(player, tmp_path):
    """--pipelined (XRSLAMAmdSetThreading(1)): the same trajectory as the ctypes session in that mode; a run is reproducible."""
    from xrslam_amd.harness import euroc, runner, scene
    seq = scene.make_sequence(n_frames=70, seed=5)
    root = euroc.write_euroc(seq, str(tmp_path / "mav0"))
    rows = []
    for k in range(2):
        tum = str(tmp_path / ("traj%d.tum" % k))
        res = _run([player, "-sc", SLAM, "-dc", SENSOR, "--tum", tum, "--no-undistort", "--pipelined", "--bootstrap-frames", "60",
                    "euroc://" + root])
        assert res["error"] == "" and res["frames"] == 70 and res["tracked"] >= 25
        assert 0 <= res["ate_rmse_m"] < 0.03
        rows.append(np.loadtxt(tum))
    np.testing.assert_array_equal(rows[0], rows[1])
    sess = runner.Session(REF_LIB, seq, slam_yaml=SLAM, sensor_yaml=SENSOR, threading=1)
    for _ in range(70):
        sess.step()
    sess.sync()
    ref = np.array([ps for ps in sess.poses if abs(ps[4]) + abs(ps[5]) + abs(ps[6]) + abs(ps[7]) > 0])
    sess.close()
    n = min(len(ref), len(rows[0]))
    assert n >= 25
    assert np.abs(ref[-n:, 1:4] - rows[0][-n:, 1:4]).max() < 5e-3


def test_cpu_reference_player_rectifies_and_stops_at_max_frames(player, tmp_path):
    from xrslam_amd.harness import euroc, scene
    dist = (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05)
    seq = scene.make_sequence(n_frames=60, seed=4, dist=dist)
    root = euroc.write_euroc(seq, str(tmp_path / "mav0"))
    res = _run([player, "--slam", SLAM, "--device", SENSOR, "--euroc", root, "--bootstrap-frames", "60", "--max-frames", "55"])
    assert res["error"] == "" and res["frames"] == 55 and res["tracked"] >= 15
    assert 0 <= res["ate_rmse_m"] < 0.03
    assert res["io_ms_per_frame"] > 0                 # PNG decode + undistortion are accounted separately


def test_cpu_reference_player_tum_reader_rectifies_a_fisheye_stream(player, tmp_path):
    """`tum://` (IO/tum_dataset_reader.cpp): same ASL layout, frames rectified with the equidistant model of
    xrslam::extra::ImageUndistorter.  The synthetic camera records through that fisheye model; rectified, the stream
    tracks like a pinhole one, unrectified it must do clearly worse."""
    from xrslam_amd.harness import euroc, scene
    coeffs = (0.0034823894022493434, 0.0007150348452162257, -0.0020532361418706202, 0.00020293673591811182)   # TUM-VI cam0
    seq = scene.make_sequence(n_frames=60, seed=6, dist=("equidistant",) + coeffs)
    root = euroc.write_euroc(seq, str(tmp_path / "mav0"))
    sensor = tmp_path / "fisheye_sensor.yaml"
    text = open(SENSOR).read()
    lines = text.splitlines()
    i = next(n for n, ln in enumerate(lines) if ln.strip().startswith("distortion:"))
    lines[i] = lines[i][:lines[i].index("distortion:")] + "distortion: [%r, %r, %r, %r]" % coeffs
    sensor.write_text("\n".join(lines) + "\n")
    base = [player, "-sc", SLAM, "-dc", str(sensor), "--bootstrap-frames", "60"]
    res = _run(base + ["tum://" + root])
    assert res["error"] == "" and res["frames"] == 60 and res["tracked"] >= 15
    assert 0 <= res["ate_rmse_m"] < 0.03
    res2 = _run(base + ["--no-undistort", "tum://" + root])
    assert not (0 <= res2["ate_rmse_m"] < 2 * res["ate_rmse_m"])

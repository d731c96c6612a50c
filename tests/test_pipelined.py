"""Pipelined mode (XRSLAMAmdSetThreading(1)): the sliding-window tracker of frame t on a thread of the library beside the
feature tracker of frame t+1 -- the reference's XRSLAM_ENABLE_THREADING build (utility/worker.h:16-60) with fixed hand-offs.

CPU (host pipeline over the oracle): the mode is reproducible bit for bit (two threads, no race decides anything), makes the
same number of frames / solves as the inline mode, tracks the synthetic stream as accurately, and can be switched off and on
between frames.  GPU: the product library in the same mode makes the same discrete decisions and produces the same poses as
the CPU pipeline in that mode (the same comparison tests/test_pipeline.py makes for the inline mode)."""
import os
import threading

import numpy as np
import pytest

from xrslam_amd.harness import runner, scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_LIB = os.path.join(ROOT, "oracle", "_build", "libxrslam_oracle.so")
N_FRAMES = 64


@pytest.fixture(scope="module")
def seq():
    return scene.make_sequence(n_frames=N_FRAMES, seed=1)


@pytest.fixture(scope="module", autouse=True)
def _oracle_built():
    if not os.path.exists(ORACLE_LIB):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


def _run(lib_path, seq, mode, slam_yaml=None, instance=False, switch_at=None):
    kw = dict(threading=mode, instance=instance)
    if slam_yaml:
        kw["slam_yaml"] = slam_yaml
    s = runner.Session(lib_path, seq, **kw)
    k = 0
    while s.step():
        assert not s.error(), s.error()
        k += 1
        if switch_at and k in switch_at:
            s.api.set_threading(switch_at[k])
    s.flush()
    s.sync()
    assert not s.error(), s.error()
    t = s.times()
    counts = (t.frames, t.solves, t.solve_iterations, t.marginalizations, t.keyframes)
    poses = np.array(s.poses)
    s.close()
    return poses, counts


@pytest.fixture(scope="module")
def cpu_pipelined(seq):
    return _run(ORACLE_LIB, seq, 1)


def test_pipelined_is_reproducible_and_accurate(seq, cpu_pipelined):
    poses, counts = cpu_pipelined
    poses2, counts2 = _run(ORACLE_LIB, seq, 1)
    assert counts2 == counts
    np.testing.assert_array_equal(poses2, poses)            # bit for bit: the hand-offs, not the scheduler, order everything
    frames, solves, iters, margs, kfs = counts
    assert frames == N_FRAMES
    assert solves >= 2 * (N_FRAMES - 38) and kfs >= 3 and margs >= 1
    assert runner.ate_rmse(list(poses), seq) < 0.03
    ok = poses[np.abs(poses[:, 4:8]).sum(1) > 0]
    assert np.allclose(np.linalg.norm(ok[:, 4:8], axis=1), 1.0, atol=1e-9)


def test_pipelined_is_the_inline_run_one_frame_late(seq, cpu_pipelined):
    """Same frames, keyframes and marginalisations as the inline run on this stream; the poses answered per frame differ
    (the state behind them is one frame older) but stay within the tracker's accuracy of each other."""
    poses_p, counts_p = cpu_pipelined
    poses_i, counts_i = _run(ORACLE_LIB, seq, 0)
    assert counts_p[0] == counts_i[0]
    assert abs(counts_p[4] - counts_i[4]) <= 1 and abs(counts_p[1] - counts_i[1]) <= 4
    ti = {round(r[0], 6): r for r in poses_i if np.abs(r[4:8]).sum() > 0}
    common = [(r, ti[round(r[0], 6)]) for r in poses_p if np.abs(r[4:8]).sum() > 0 and round(r[0], 6) in ti]
    assert len(common) >= N_FRAMES - 42
    d = np.array([np.linalg.norm(a[1:4] - b[1:4]) for a, b in common])
    assert d.max() < 0.05
    assert not np.array_equal(poses_p, poses_i)             # it IS another schedule: never compare across modes bit for bit


def test_switching_modes_between_frames(seq):
    """Inline for 45 frames, pipelined for 10, inline again: every frame is processed exactly once, nothing is lost at the
    switches (set_threading waits for the job in flight)."""
    poses, counts = _run(ORACLE_LIB, seq, 0, switch_at={45: 1, 55: 0})
    assert counts[0] == N_FRAMES
    assert runner.ate_rmse(list(poses), seq) < 0.03


def test_two_pipelined_instances_from_two_threads(seq):
    """Two instances, each with a backend thread of its own, driven from two Python threads: each reproduces the
    single-instance pipelined run exactly."""
    alone, counts = _run(ORACLE_LIB, seq, 1, instance=True)
    out = [None, None]

    def work(i):
        out[i] = _run(ORACLE_LIB, seq, 1, instance=True)

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for poses, c in out:
        assert c == counts
        np.testing.assert_array_equal(poses, alone)


@pytest.mark.gpu
def test_gpu_pipelined_matches_cpu_pipelined(seq, cpu_pipelined):
    from xrslam_amd import _lib
    poses_o, counts_o = cpu_pipelined
    poses_h, counts_h = _run(_lib.LIB_PATH, seq, 1)
    assert counts_h == counts_o                              # identical discrete decisions
    assert poses_h.shape == poses_o.shape
    np.testing.assert_allclose(poses_h[:, 0], poses_o[:, 0], rtol=0, atol=0)
    np.testing.assert_allclose(poses_h[:, 1:], poses_o[:, 1:], rtol=1e-4, atol=1e-6)
    assert runner.ate_rmse(list(poses_h), seq) < 0.03
    poses_h2, counts_h2 = _run(_lib.LIB_PATH, seq, 1)       # and the GPU run itself is reproducible
    assert counts_h2 == counts_h
    np.testing.assert_array_equal(poses_h2, poses_h)


@pytest.mark.gpu
def test_gpu_pipelined_matches_cpu_pipelined_stress_config():
    """BASELINE config 3 (300 features, 15-keyframe window) in pipelined mode."""
    from xrslam_amd import _lib
    yaml = os.path.join(ROOT, "configs", "stress_slam_300.yaml")
    seq = scene.make_sequence(n_frames=100, seed=3)
    poses_o, counts_o = _run(ORACLE_LIB, seq, 1, yaml)
    poses_h, counts_h = _run(_lib.LIB_PATH, seq, 1, yaml)
    assert counts_h == counts_o
    assert counts_o[4] >= 8 and counts_o[3] >= 1
    assert poses_h.shape == poses_o.shape
    np.testing.assert_allclose(poses_h[:, 1:], poses_o[:, 1:], rtol=1e-4, atol=1e-6)
    assert runner.ate_rmse(list(poses_h), seq) < 0.03


def test_flush_destroy_and_results_with_a_frame_in_flight(seq):
    """XRSLAMAmdFlush on an idle pipeline is a no-op; LANDMARKS / BIAS results and the statistics getters wait for the backend
    job in flight by themselves; Destroy with a job in flight joins it (no crash, no leak of the worker thread)."""
    import ctypes as C
    s = runner.Session(ORACLE_LIB, seq, threading=1)
    s.sync()                                     # nothing in flight yet
    for _ in range(50):
        assert s.step()
    bias = (C.c_double * 6)()
    s.api.get_result(5, C.byref(bias))           # XRSLAM_RESULT_BIAS: joins the job of the frame just posted
    t = s.times()                                # ... and so do the statistics
    assert t.frames >= 49 and t.solves > 0
    assert s.step() and s.step()
    s.close()                                    # Destroy while the backend is (most likely) still working on the last frame
    s2 = runner.Session(ORACLE_LIB, seq, threading=1)     # the process-global instance is reusable afterwards
    for _ in range(45):
        assert s2.step()
    s2.sync()
    assert not s2.error(), s2.error()
    s2.close()


def test_pipelined_request_with_rdvio_filter_stays_inline(tmp_path):
    """With parsac.parsac_flag the backend reads the tracking map (update_track_status): threading mode 1 is accepted but the
    frames run inline -- the trajectory is the inline one, bit for bit."""
    text = open(os.path.join(ROOT, "configs", "euroc_slam.yaml")).read()
    assert "parsac_flag" in text
    import re
    yaml_on = tmp_path / "parsac_slam.yaml"
    yaml_on.write_text(re.sub(r"parsac_flag:\s*\S+", "parsac_flag: true", text))
    sq = scene.make_sequence(n_frames=56, seed=2)
    inline, c_inline = _run(ORACLE_LIB, sq, 0, str(yaml_on))
    asked, c_asked = _run(ORACLE_LIB, sq, 1, str(yaml_on))
    assert c_asked == c_inline
    np.testing.assert_array_equal(asked, inline)

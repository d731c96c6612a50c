"""The initialiser against an independent model (row f3).

Until round 3 the initialiser's only parity evidence was "the GPU library and the CPU reference pipeline agree" -- two builds of the
SAME host source.  Here the C++ pipeline logs what every decision of every initialisation attempt looked at
(XRSLAM_AMD_DUMP_INIT), and tests/init_model.py -- numpy, written from /root/reference/xrslam/src/xrslam/core/initializer.cpp:22-571,
sharing no code with csrc/host/pipeline.hpp / two_view.hpp -- re-derives every answer from those inputs: the key frames picked and
the IMU samples they carry, that the eight (R, T) hypotheses are the decompositions of the logged H and E in the reference's order,
the two-view triangulations and the (quirky) vote, the gyroscope bias, gravity / scale / velocities, the tangent-plane refinement,
the scale gates and apply_init's rotation.  The stream is the self-initialising one of tests/test_pipeline.py (several failed
attempts before the one that succeeds: both outcomes of the gates are exercised); the CPU run needs no GPU, the GPU run repeats the
comparison on the product library's log."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from tests import init_model as im
from xrslam_amd.harness import runner, scene
from xrslam_amd.harness.trajectory import Trajectory

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_LIB = os.path.join(ROOT, "oracle", "_build", "libxrslam_oracle.so")
N_FRAMES = 80


@pytest.fixture(scope="module")
def init_seq():
    if not os.path.exists(ORACLE_LIB):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    return scene.make_sequence(n_frames=N_FRAMES, seed=1, traj=Trajectory(amp=1.5, speed=0.3))


def _run_logged(lib_path, seq):
    fd, path = tempfile.mkstemp(prefix="xr_init_", suffix=".jsonl")
    os.close(fd)
    os.environ["XRSLAM_AMD_DUMP_INIT"] = path
    try:
        s = runner.Session(lib_path, seq, init_frames=0)
    finally:
        del os.environ["XRSLAM_AMD_DUMP_INIT"]
    while s.step():
        assert not s.error(), s.error()
    s.flush()
    rep = s.init_report()
    out = (rep.attempts, rep.successes, rep.sfm_candidate, rep.scale)
    s.close()
    recs = im.read(path)
    os.unlink(path)
    return recs, out


def _check_log(recs, report):
    attempts, successes, candidate, scale = report
    by = {}
    for r in recs:
        by.setdefault(r["what"], []).append(r)
    # ---- mirror_keyframe_map: every call that built a window
    assert len(by["keyframes"]) >= attempts
    for r in by["keyframes"]:
        want = im.keyframes(r)
        assert want is not None
        picked, samples, t0, t1 = want
        assert [int(v) for v in r["picked"]] == picked
        assert len(picked) == 8 and picked[-1] == int(r["init_frame_id"]) and picked[1] - picked[0] == 5
        assert [int(v) for v in r["picked_samples"]][1:] == samples[1:]
        np.testing.assert_array_equal(np.array(r["picked_t0"])[1:], np.array(t0)[1:])
        np.testing.assert_array_equal(np.array(r["picked_t1"])[1:], np.array(t1)[1:])
        assert all(v >= 50 for v in r["links_to_next"][:-1])          # consecutive key frames share tracks
    # ---- init_sfm: one vote per attempt that passed the match / parallax gates
    votes = by["sfm_vote"]
    assert len(votes) == attempts >= 2
    won = 0
    for r in votes:
        assert len(r["pi"]) // 2 >= int(r["min_matches"]) and r["parallax"] >= r["min_parallax"]
        d = im.two_view_models(r)
        assert d["rotation_defect"] < 1e-9 and d["unit_T_defect"] < 1e-12 and d["pairing_defect"] == 0.0, d
        assert d["essential_defect"] < 1e-8, d
        assert d["homography_defect"] < 1e-6, d
        assert d["essential_inliers"] > 0.9, d
        counts, scores, best, pts, status = im.triangulation_vote(r)
        assert counts == [int(v) for v in r["counts"]]
        np.testing.assert_allclose(scores, r["scores"], rtol=1e-6, atol=1e-12)
        assert best == int(r["best"])
        np.testing.assert_array_equal(status, np.array(r["status_best"], int))
        np.testing.assert_allclose(pts[status == 1], np.array(r["points_best"]).reshape(-1, 3)[status == 1], rtol=1e-7, atol=1e-9)
        won += counts[best] >= int(r["min_triangulation"])
    assert int(votes[-1]["best"]) == candidate
    # ---- init_imu: every attempt whose SfM succeeded
    assert len(by["gyro_bias"]) == len(by["gravity_scale_velocity"]) == len(by["imu_result"]) <= won
    for r in by["gyro_bias"]:
        np.testing.assert_allclose(im.gyro_bias(r), r["bg"], rtol=1e-7, atol=1e-12)
    for r in by["gravity_scale_velocity"]:
        g, s, v = im.gravity_scale_velocity(r)
        np.testing.assert_allclose(g, r["gravity"], rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(s, r["scale"], rtol=1e-7)
        np.testing.assert_allclose(v.ravel(), r["velocities"], rtol=1e-6, atol=1e-9)
        assert abs(np.linalg.norm(r["gravity"]) - im.GRAVITY_NOMINAL) < 1e-9
    for r in by.get("refine_via_gravity", []):
        g, s, v = im.refine_via_gravity(r)
        np.testing.assert_allclose(g, r["gravity"], rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(s, r["scale"], rtol=1e-7)
        np.testing.assert_allclose(v.ravel(), r["velocities"], rtol=1e-6, atol=1e-9)
    # the scale gates (initializer.cpp:391, 397): an attempt fails exactly when a solve left the scale outside [0.001, 1] -- or too
    # few landmarks survive apply_init
    gsv, ref, app = by["gravity_scale_velocity"], by.get("refine_via_gravity", []), by.get("apply_init", [])
    ri = ai = 0
    outcomes = []
    for k, res in enumerate(by["imu_result"]):
        ok = 0.001 <= gsv[k]["scale"] <= 1.0
        if ok and res["refine_imu"]:
            ok = 0.001 <= ref[ri]["scale"] <= 1.0
            ri += 1
        if ok:
            ok = res["final_points"] >= res["min_landmarks"]
            ai += 1
        assert bool(res["ok"]) == ok
        outcomes.append(ok)
    assert ri == len(ref) and ai == len(app)
    assert sum(outcomes) == successes == 1 and not all(outcomes[:-1] + [False])     # failures before the success: both branches ran
    # ---- apply_init
    for r in app:
        poses, vs = im.apply_init(r)
        got = np.array(r["imu_pose_after"]).reshape(-1, 7)
        sign = np.sign((poses[:, :4] * got[:, :4]).sum(1))[:, None]      # q and -q are the same rotation
        np.testing.assert_allclose(poses[:, :4] * sign, got[:, :4], rtol=0, atol=1e-9)
        np.testing.assert_allclose(poses[:, 4:], got[:, 4:], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(vs.ravel(), r["v_after"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(app[-1]["scale"], scale, rtol=1e-12)


def test_initialiser_decisions_match_the_independent_model_cpu(init_seq):
    recs, report = _run_logged(ORACLE_LIB, init_seq)
    _check_log(recs, report)


@pytest.mark.gpu
def test_initialiser_decisions_match_the_independent_model_gpu(init_seq):
    from xrslam_amd import _lib
    recs, report = _run_logged(_lib.LIB_PATH, init_seq)
    _check_log(recs, report)

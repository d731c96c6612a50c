"""The RD-VIO filters against an independent model (row f4).

The dynamic-object stream of tests/test_pipeline.py (a textured box swinging through the view, parsac.parsac_flag on) runs through the
C++ pipeline with the decision log on (XRSLAM_AMD_DUMP_INIT); tests/parsac_model.py -- numpy, written from
/root/reference/xrslam/src/xrslam/utility/parsac.h, imu_parsac.h and sliding_window_tracker.cpp:577-739 -- replays every PARSAC run
from its logged inputs and per-hypothesis inlier masks and must arrive at the same bucketing, prior, scores, best-hypothesis
hand-overs, iteration caps, iteration counts, masks, give-ups and written-back bin confidences; and at judge_track_status's verdicts
from the logged epipolar distances.  Also checked: the bin confidences one run writes are the prior the next run of its kind reads
(function-local statics in the reference, geometry/stereo.cpp:149 -- tracker state here)."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from tests import init_model as im
from tests import parsac_model as pm
from xrslam_amd.harness import runner, scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_LIB = os.path.join(ROOT, "oracle", "_build", "libxrslam_oracle.so")
RDVIO_YAML = os.path.join(ROOT, "configs", "rdvio_slam_150.yaml")


def _moving_object(t):
    return np.array([-1.4 + 0.5 * np.sin(1.2 * t), 0.35 * np.cos(0.9 * t), 2.0]), np.array([0.45, 0.45, 0.2])


@pytest.fixture(scope="module")
def dynamic_seq():
    if not os.path.exists(ORACLE_LIB):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    return scene.make_sequence(n_frames=100, seed=1, moving_object=_moving_object)


def _run_logged(lib_path, seq):
    fd, path = tempfile.mkstemp(prefix="xr_rd_", suffix=".jsonl")
    os.close(fd)
    os.environ["XRSLAM_AMD_DUMP_INIT"] = path
    try:
        s = runner.Session(lib_path, seq, slam_yaml=RDVIO_YAML)
    finally:
        del os.environ["XRSLAM_AMD_DUMP_INIT"]
    while s.step():
        assert not s.error(), s.error()
    s.flush()
    t = s.times()
    counters = (t.wall_scope[12], t.wall_scope[13])
    s.close()
    recs = im.read(path)
    os.unlink(path)
    return recs, counters


def _check(recs, counters):
    runs = [r for r in recs if r["what"] == "parsac_run"]
    judged = [r for r in recs if r["what"] == "judge_track_status"]
    pnp = [r for r in runs if r["imu"]]
    ess = [r for r in runs if not r["imu"]]
    assert len(pnp) >= 40 and len(ess) >= 1 and len(judged) >= 40
    takeovers = giveups = capped = 0
    for r in runs:
        m = pm.replay(r)
        assert m["nvalid"] == int(r["nvalid"])
        np.testing.assert_array_equal(m["accumulated_prior"], np.array(r["accumulated_prior"], np.float32))
        np.testing.assert_allclose(m["scores"], r["cand_score"], rtol=2e-6, atol=1e-12)
        assert m["takes_over"] == [bool(v) for v in r["cand_takes_over"]]
        assert m["iter_max_after"] == [int(v) for v in r["cand_iter_max_after"]]
        assert m["gave_up"] == bool(r["gave_up"])
        assert m["inlier_count"] == int(r["inlier_count"])
        # the loop runs until the (adaptive) cap the last take-over left (that iteration itself always completes)
        cap = m["iter_max_after"][-1] if m["iter_max_after"] else int(r["max_iteration"])
        assert int(r["iterations_run"]) == m["iterations_run"]
        if not m["gave_up"]:
            if m["final_mask"] is not None:
                np.testing.assert_array_equal(m["final_mask"], np.array(r["final_mask"], int))
            np.testing.assert_allclose(m["bins_after"], np.array(r["bins_after"], np.float32), rtol=2e-6, atol=0)
        else:
            np.testing.assert_array_equal(np.array(r["bins_after"]), np.array(r["bins_before"]))     # a give-up writes nothing back
        takeovers += sum(m["takes_over"])
        giveups += m["gave_up"]
        capped += cap < int(r["max_iteration"])
    assert takeovers >= len(runs) and capped >= len(runs) // 2      # hypotheses did compete, and the adaptive cap did cut runs short
    # one run's written-back confidences are the next run's prior, per kind
    for group in (pnp, ess):
        for a, b in zip(group, group[1:]):
            np.testing.assert_array_equal(np.array(a["bins_after"]), np.array(b["bins_before"]))
    assert np.array(pnp[0]["bins_before"]).tolist() == [0.5] * 400
    # judge_track_status: separation verdict and the 2D-2D threshold it hands on
    sep = 0
    for r in judged:
        ok, thr = pm.judge(r)
        assert ok == bool(r["separated"])
        if ok:
            np.testing.assert_allclose(thr, r["threshold"], rtol=1e-12)
        sep += ok
    assert sep == counters[0] >= 1                   # XRSLAMAmdTimes.wall_scope[12]: frames on which a dynamic group was separated
    # every 2D-2D run follows a separated verdict and uses its threshold (sliding_window_tracker.cpp:523-547: th / fx)
    assert len(ess) >= sep


def test_parsac_decisions_match_the_independent_model_cpu(dynamic_seq):
    _check(*_run_logged(ORACLE_LIB, dynamic_seq))


@pytest.mark.gpu
def test_parsac_decisions_match_the_independent_model_gpu(dynamic_seq):
    from xrslam_amd import _lib
    _check(*_run_logged(_lib.LIB_PATH, dynamic_seq))

"""The C++ sliding-window tracker against an independent Python model of the reference's structural decisions.

The host pipeline (xrslam_amd/csrc/host/pipeline.hpp) restates core/sliding_window_tracker.cpp function by function, and
until now was only compared with itself (GPU build vs CPU build of the same source).  Here every frame's inputs to
manage_keyframe -- new frame id, its FT_NO_TRANSLATION tag, the number of mapped landmarks in view -- are logged by the C++
tracker (XRSLAM_AMD_DUMP_SWT) and replayed through tests/swt_model.py, a 60-line model written from the reference
(:145-223, :360-393); the two must agree on keyframe / subframe at every frame and on the whole window afterwards (frame ids,
tags, subframe lists, the merging of rotation-only subframe triples, the keyframe that slides out).  The stream stops
translating for a while (pure rotation), so the lift / re-attach branches and the triple merge are exercised, not only the
landmark-count rule.  CPU: the pipeline over the oracle; the GPU build runs the same host source."""
import json
import os

import numpy as np
import pytest

from tests.swt_model import SwtModel
from xrslam_amd.harness import runner, scene
from xrslam_amd.harness.trajectory import Trajectory

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_LIB = os.path.join(ROOT, "oracle", "_build", "libxrslam_oracle.so")
SLAM = os.path.join(ROOT, "configs", "bench_slam_150.yaml")


class PausingTrajectory(Trajectory):
    """The S1 figure-eight whose translation freezes between t0 and t1 (smoothly) while the attitude keeps moving."""

    def __init__(self, t0, t1, **kw):
        super().__init__(**kw)
        self.t0, self.t1 = t0, t1

    def _warp(self, t):
        # position runs on a warped clock g(t): g' = 1 outside [t0 - 0.5, t1 + 0.5], 0 inside [t0, t1], C1 in between
        a, b = self.t0, self.t1

        def ramp(x):   # integral of a smoothstep falling from 1 to 0 over a unit interval
            x = np.clip(x, 0.0, 1.0)
            return x - (x ** 3 - 0.5 * x ** 4)
        g = t if t < a - 0.5 else (a - 0.5) + 0.5 * ramp((t - (a - 0.5)) / 0.5) if t < a else None
        if g is None:
            held = (a - 0.5) + 0.5 * ramp(1.0)
            if t <= b:
                g = held
            elif t < b + 0.5:
                x = (t - b) / 0.5
                g = held + 0.5 * (np.clip(x, 0, 1) ** 3 - 0.5 * np.clip(x, 0, 1) ** 4)
            else:
                g = held + 0.5 * 0.5 + (t - (b + 0.5))
        return g

    def p(self, t):
        return super().p(self._warp(t))


def _config_values():
    import yaml
    with open(SLAM) as fh:
        y = yaml.safe_load(fh.read().replace("%YAML:1.0", "", 1))   # the OpenCV FileStorage directive the reference's files carry
    sw = y["sliding_window"]
    return int(sw["size"]), int(sw["subframe_size"]), int(sw["force_keyframe_landmarks"])


def test_tracker_decisions_match_the_reference_model(tmp_path):
    if not os.path.exists(ORACLE_LIB):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    log = str(tmp_path / "swt.jsonl")
    os.environ["XRSLAM_AMD_DUMP_SWT"] = log
    try:
        seq = scene.make_sequence(n_frames=170, seed=1, traj=PausingTrajectory(9.0, 10.6))
        s = runner.Session(ORACLE_LIB, seq, slam_yaml=SLAM)
        while s.step():
            assert not s.error(), s.error()
        s.flush()
        n_pose = len(s.poses)
        s.close()
    finally:
        del os.environ["XRSLAM_AMD_DUMP_SWT"]
    rows = [r for r in (json.loads(ln) for ln in open(log)) if "window" in r]      # (the log also carries mirror_frame's records)
    assert len(rows) >= 120 and n_pose >= 120
    size, subframe_size, force = _config_values()
    # the model starts from the window the C++ tracker reported after its first frame and replays everything after it
    model = SwtModel(rows[0]["window"], size, subframe_size, force)
    seen = dict(keyframe=0, subframe=0, lift_to_keyframe=0, hang_below_lifted=0, merged=0, rotation_frames=0)
    for k, r in enumerate(rows[1:], start=1):
        before = json.dumps(model.window)
        n_sub_before = sum(len(f[2]) for f in model.window)
        is_kf = model.step(r["frame"], r["no_translation"], r["mapped"])
        assert is_kf == bool(r["keyframe"]), "frame %d (log line %d): keyframe decision differs; window before: %s" % (r["frame"], k, before)
        assert model.window == r["window"], "frame %d (log line %d): window differs\nmodel %s\nC++   %s" % (r["frame"], k, model.window, r["window"])
        seen["keyframe" if is_kf else "subframe"] += 1
        seen["rotation_frames"] += r["no_translation"]
        if is_kf and model.window[-1][0] != r["frame"]:
            if model.window[-1][2] and model.window[-1][2][-1][0] == r["frame"]:
                seen["hang_below_lifted"] += 1
        if is_kf and len(model.window) >= 2 and model.window[-1][0] == r["frame"] and model.window[-2][0] != json.loads(before)[-1][0]:
            seen["lift_to_keyframe"] += 1
        if not is_kf and sum(len(f[2]) for f in model.window) < n_sub_before + 1:
            seen["merged"] += 1
    # the stream must have exercised the branches this test is about
    assert seen["keyframe"] >= 15 and seen["subframe"] >= 40, seen
    assert seen["rotation_frames"] >= 10, seen
    assert seen["hang_below_lifted"] >= 1 and seen["lift_to_keyframe"] >= 1, seen


def test_mirror_frame_links_match_an_independent_model(tmp_path):
    """mirror_frame (core/sliding_window_tracker.cpp:31-80), so far only compared with itself.  The C++ pipeline logs the
    tracking map's side -- for the last two frames, which track every keypoint is on -- and, per mirrored frame, the links it
    made: (keypoint of the last window frame, keypoint of the new frame, window-map track, created or continued) plus what the
    new frame carries after the prune.  The model below derives the links from the tracking map's structure alone and keeps the
    newest window frame's keypoint -> track table itself: same links in the same order, a track is continued exactly when the
    model knows one on that keypoint (and then it is that track), created otherwise, and the table after the prune only ever
    loses entries."""
    if not os.path.exists(ORACLE_LIB):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    log = str(tmp_path / "swt.jsonl")
    os.environ["XRSLAM_AMD_DUMP_SWT"] = log
    try:
        seq = scene.make_sequence(n_frames=110, seed=4)
        s = runner.Session(ORACLE_LIB, seq, slam_yaml=SLAM)
        while s.step():
            assert not s.error(), s.error()
        s.flush()
        s.close()
    finally:
        del os.environ["XRSLAM_AMD_DUMP_SWT"]
    ft = {}                      # tracking map: frame id -> track id per keypoint (-1: none), latest record wins
    newest = None                # (frame id, {keypoint: window-map track}) of the newest window frame, as the model knows it
    n_mirror = n_links = n_created = n_continued = n_pruned = 0
    seen_ids = set()
    for ln in open(log):
        r = json.loads(ln)
        if "ft" in r:
            for f in r["ft"]:
                ft[f["id"]] = f["tracks"]
        elif "mirror" in r:
            i, j = r["from"], r["mirror"]
            assert i in ft and j in ft
            pos_j = {t: k for k, t in enumerate(ft[j]) if t != -1}
            expect = [(ki, pos_j[t]) for ki, t in enumerate(ft[i]) if t != -1 and t in pos_j]
            got = [(a, b) for a, b, _, _ in r["links"]]
            assert got == expect, "frame %d: links differ" % j
            table = newest[1] if newest is not None and newest[0] == i else None
            after = {}
            for ki, kj, tid, created in r["links"]:
                if table is not None:
                    if ki in table:
                        assert not created and tid == table[ki], "frame %d keypoint %d: a known track was not continued" % (j, ki)
                    else:
                        assert created, "frame %d keypoint %d: continued a track the model does not know" % (j, ki)
                if created:
                    assert tid not in seen_ids                       # a fresh id
                    n_created += 1
                else:
                    n_continued += 1
                seen_ids.add(tid)
                assert kj not in after                                # one track per keypoint
                after[kj] = tid
            logged_after = {k: t for k, t in r["after"]}
            assert set(logged_after.items()) <= set(after.items())   # the prune only removes
            n_pruned += len(after) - len(logged_after)
            newest = (j, logged_after)
            n_mirror += 1
            n_links += len(got)
    assert n_mirror >= 60 and n_links >= 60 * 80, (n_mirror, n_links)
    assert n_created >= 150 and n_continued >= 40 * 80, (n_created, n_continued)


def test_landmark_decisions_match_an_independent_model(tmp_path):
    """track_landmark and the landmark sweep of refine_window (core/sliding_window_tracker.cpp:225-245, 323-357; map/track.cpp:46-101),
    so far read but not tested.  The C++ pipeline logs, for every track it triangulates or sweeps, the
    observations the decision looks at and what it decided; tests/swt_model.py restates both decisions from the reference with
    numpy.  Same accept / reject for every triangulation, the same anchored inverse depth, the same validity verdict for every
    landmark after every window solve (the stream's own tracking errors and short baselines produce the rejections)."""
    if not os.path.exists(ORACLE_LIB):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    from tests import swt_model as sm
    log = str(tmp_path / "swt.jsonl")
    os.environ["XRSLAM_AMD_DUMP_SWT"] = log
    try:
        seq = scene.make_sequence(n_frames=90, seed=2)
        s = runner.Session(ORACLE_LIB, seq, slam_yaml=SLAM)
        while s.step():
            assert not s.error(), s.error()
        s.flush()
        s.close()
    finally:
        del os.environ["XRSLAM_AMD_DUMP_SWT"]
    n_tri = n_tri_rejected = n_cull = n_invalid = n_untri = 0
    for ln in open(log):
        r = json.loads(ln)
        if "triangulate" in r:
            obs = r["obs"]
            p = sm.triangulate(obs)
            assert (p is not None) == bool(r["ok"]), "track %d at frame %d: triangulation verdict differs" % (r["track"], r["triangulate"])
            if p is None:
                assert r["inv_depth"] == -1.0
                n_tri_rejected += 1
            else:
                np.testing.assert_allclose(sm.anchor_inv_depth(obs, p), r["inv_depth"], rtol=1e-7)
            n_tri += 1
        elif "cull" in r:
            if r["triangulated"]:
                valid = sm.landmark_is_valid(r["obs"], r["inv_depth"])
                assert valid == bool(r["valid"]), "track %d after the solve of frame %d: validity differs" % (r["track"], r["cull"])
                assert r["inv_depth_after"] == r["inv_depth"]
                n_invalid += not valid
                n_cull += 1
            else:
                assert r["inv_depth_after"] == -1.0      # not triangulated: the inverse depth is reset (:351-353)
                n_untri += 1
    print("triangulations %d (%d rejected), swept landmarks %d (%d invalid), untriangulated %d" % (n_tri, n_tri_rejected, n_cull, n_invalid, n_untri))
    assert n_tri >= 200 and n_cull >= 2000, (n_tri, n_cull)
    assert n_tri_rejected + n_invalid >= 1, (n_tri_rejected, n_invalid)   # the stream exercised a rejection somewhere


@pytest.fixture(scope="module")
def logged_run(tmp_path_factory):
    """One 120-frame run of the CPU reference pipeline with the decision log AND the solver-problem dump on (shared by the three
    problem-assembly tests below): -> (path of the log, directory of the dumps)."""
    if not os.path.exists(ORACLE_LIB):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    base = tmp_path_factory.mktemp("logged_run")
    log = str(base / "swt.jsonl")
    dump_dir = base / "ba"
    dump_dir.mkdir()
    os.environ["XRSLAM_AMD_DUMP_SWT"] = log
    os.environ["XRSLAM_AMD_DUMP_BA"] = str(dump_dir)
    try:
        seq = scene.make_sequence(n_frames=120, seed=2)
        s = runner.Session(ORACLE_LIB, seq, slam_yaml=SLAM)
        while s.step():
            assert not s.error(), s.error()
        s.flush()
        s.close()
    finally:
        del os.environ["XRSLAM_AMD_DUMP_SWT"]
        del os.environ["XRSLAM_AMD_DUMP_BA"]
    return log, str(dump_dir)


def test_localize_problem_is_what_an_independent_track_model_predicts(logged_run):
    """localize_newframe's problem assembly (core/sliding_window_tracker.cpp:119-143), so far read but not tested (VERDICT r2, weak
    #14): ONE free frame (the new one), its predecessor and the landmarks' reference keyframes constant, one pre-integration prior
    factor, and one reprojection prior factor per key point of the new frame whose track is valid AND triangulated -- depths constant.  The expected
    factor count comes from a model that never looks at the tracker's tags: it keeps every window-map track's (triangulated, valid)
    pair from the landmark-sweep records after each window solve (the verdicts test_landmark_decisions_... re-derives with numpy) and
    reads which track each key point of the new frame is on from mirror_frame's record.  Compared with (a) the count the tracker
    logs as manage_keyframe's input and (b) the problem the solver actually received (XRSLAM_AMD_DUMP_BA)."""
    import glob
    from tests import ba_snapshots as snap
    log, dump_dir = logged_run
    # ---- the track model over the log
    state = {}                 # window-map track id -> (triangulated, valid), as of the last landmark sweep that saw it
    have_snapshot = False
    expected = {}              # frame id -> predicted number of reprojection prior factors
    logged = {}
    order = []
    for ln in open(log):
        r = json.loads(ln)
        if "mirror" in r:
            if have_snapshot:
                expected[r["mirror"]] = sum(1 for _kj, tid in r["after"] if state.get(tid, (0, 0)) == (1, 1))
        elif "cull" in r:
            state[r["track"]] = (1 if r["triangulated"] else 0, 1 if r["valid"] else 0)
            have_snapshot = True
        elif "window" in r:
            logged[r["frame"]] = r["mapped"]
            order.append(r["frame"])
    checked = [f for f in order if f in expected]
    assert len(checked) >= 60, len(checked)
    for f in checked:
        assert expected[f] == logged[f], "frame %d: model predicts %d mapped key points, the tracker counted %d" % (f, expected[f], logged[f])
    # ---- the problems the solver received: two solves per tracked frame, the first one is localize_newframe's
    by_count = {}
    for path in sorted(glob.glob(os.path.join(str(dump_dir), "*.xrba"))):
        fc = int(os.path.basename(path).split("_f")[1].split("_")[0])
        by_count.setdefault(fc, []).append(path)
    groups = [by_count[k] for k in sorted(by_count) if len(by_count[k]) == 2]
    assert len(groups) >= len(order) - 1                   # (the initialiser's own solves come in other group sizes)
    groups = groups[-len(order):]
    n_checked = 0
    for fid, (first, _second) in zip(order[-len(groups):], groups):
        d = snap.read_xrba(first)
        free = np.flatnonzero(d["frame_fix"] != 3)
        assert len(free) == 1, first                                          # the new frame; its predecessor and the landmarks'
        assert d["frame_fix"][free[0]] == 0                                   # reference keyframes enter as constants
        assert np.all(d["obs_ref"] != free[0])
        assert len(d["imu_i"]) == 1 and d["imu_j"][0] == free[0] and d["imu_i"][0] != free[0]
        assert len(d["rot_tgt"]) == 0 and len(d["prior_frames"]) == 0
        assert np.all(d["landmark_fix"] == 1)                                # prior factors: depths are constants
        assert np.all(d["obs_tgt"] == free[0])
        assert len(np.unique(d["obs_lm"])) == len(d["obs_lm"])               # one factor per landmark
        if fid in expected:
            assert len(d["obs_tgt"]) == expected[fid], "frame %d: %d reprojection prior factors, model predicts %d" % (
                fid, len(d["obs_tgt"]), expected[fid])
            n_checked += 1
    assert n_checked >= 60, n_checked


def test_window_problem_is_what_an_independent_track_model_predicts(logged_run):
    """refine_window's problem assembly (core/sliding_window_tracker.cpp:247-322): every keyframe of the window free, one
    pre-integration factor between consecutive keyframes, the marginalisation prior, a free inverse depth per valid track anchored in
    a keyframe, and one reprojection factor per observation of a valid, triangulated track in a keyframe other than its anchor.  The
    expected counts come from the same tag-free track model as above: (triangulated, valid) pairs from the sweep records of the
    previous window solves and this keyframe's triangulation records, observations and anchors from this keyframe's sweep records.
    Compared with the problem the solver received (XRSLAM_AMD_DUMP_BA)."""
    import glob
    from tests import ba_snapshots as snap
    log, dump_dir = logged_run
    state = {}            # track id -> (triangulated, valid) BEFORE the window solve being looked at
    have_snapshot = False
    pending = []          # this keyframe's sweep records
    expected = {}         # frame id -> (reprojection factors, free landmarks)
    order = []
    for ln in open(log):
        r = json.loads(ln)
        if "triangulate" in r:
            state[r["track"]] = (1, 1) if r["ok"] else (0, 0)          # track_landmark (:225-245)
        elif "cull" in r:
            pending.append(r)
        elif "window" in r:
            order.append((r["frame"], r["keyframe"]))
            if r["keyframe"]:
                assert pending, r["frame"]
                if have_snapshot:
                    m = free = 0
                    for c in pending:
                        tri, valid = state.get(c["track"], (0, 0))
                        kf = [o[10] for o in c["obs"]]
                        if not valid or not kf or kf[0] != 1:            # not a parameter: invalid, or anchored in a subframe
                            continue
                        free += 1
                        if tri:
                            m += sum(kf) - 1                             # every keyframe observation but the anchor's own
                    expected[r["frame"]] = (m, free)
                for c in pending:                                        # the sweep's verdicts are the next solves' inputs
                    state[c["track"]] = (1 if c["triangulated"] else 0, 1 if c["valid"] else 0)
                have_snapshot = True
                pending = []
            else:
                assert not pending
    by_count = {}
    for path in sorted(glob.glob(os.path.join(str(dump_dir), "*.xrba"))):
        fc = int(os.path.basename(path).split("_f")[1].split("_")[0])
        by_count.setdefault(fc, []).append(path)
    groups = [by_count[k] for k in sorted(by_count) if len(by_count[k]) == 2][-len(order):]
    assert len(groups) == len(order)
    n_checked = 0
    for (fid, is_kf), (_first, second) in zip(order, groups):
        if not is_kf or fid not in expected:
            continue
        d = snap.read_xrba(second)
        F = len(d["frame_state"])
        assert np.all(d["frame_fix"][1:] == 0) and d["frame_fix"][0] in (0, 1), second   # every keyframe free (the oldest keeps the
                                                                                       # initialiser's pose gauge until it slides out)
        assert len(d["imu_i"]) == F - 1 and np.all(d["imu_j"] - d["imu_i"] == 1)
        npf = len(d["prior_frames"])                                         # the prior covers the oldest frames: those that were in the
        assert 0 < npf <= F - 1 and np.array_equal(d["prior_frames"], np.arange(npf))   # window when it was last rebuilt
        assert len(d["rot_tgt"]) == 0
        m, free = expected[fid]
        assert int((d["landmark_fix"] == 0).sum()) == free, "frame %d: %d free landmarks, model predicts %d" % (
            fid, int((d["landmark_fix"] == 0).sum()), free)
        assert len(d["obs_tgt"]) == m, "frame %d: %d reprojection factors, model predicts %d" % (fid, len(d["obs_tgt"]), m)
        assert np.all(d["obs_tgt"] != d["obs_ref"])
        n_checked += 1
    assert n_checked >= 12, n_checked


def test_subwindow_problem_is_what_an_independent_track_model_predicts(logged_run):
    """refine_subwindow's problem assembly, translating branch (core/sliding_window_tracker.cpp:414-465): the newest keyframe constant,
    its subframes free, one pre-integration factor per subframe (chained from the keyframe), and for every subframe one reprojection
    prior factor per key point whose track is valid, triangulated and anchored in a keyframe.  Expected counts from the tag-free
    model: the subframe list from the tracker's window record, each subframe's key point -> track assignment from its own
    mirror_frame record, (triangulated, valid, anchored-in-a-keyframe) from the last landmark sweep.  Compared with the problem the
    solver received for every non-keyframe frame of a 120-frame run."""
    import glob
    from tests import ba_snapshots as snap
    log, dump_dir = logged_run
    state = {}        # track id -> [triangulated, valid, [(frame id, is keyframe) of its observations]], as of the last sweep
    after = {}        # frame id -> [(key point, track id)] when it was mirrored
    have_snapshot = False
    expected = {}     # frame id -> (reprojection prior factors, subframes)
    order = []
    prev_window = None
    for ln in open(log):
        r = json.loads(ln)
        if "mirror" in r:
            after[r["mirror"]] = [tuple(x) for x in r["after"]]
        elif "cull" in r:
            state[r["track"]] = [1 if r["triangulated"] else 0, 1 if r["valid"] else 0, [(o[15], o[10]) for o in r["obs"]]]
            have_snapshot = True
        elif "window" in r:
            order.append((r["frame"], r["keyframe"]))
            # frames that left the window since the previous record (slide_window: the oldest keyframe and its subframes) no longer
            # carry observations: a track anchored there is now anchored in its next observation (Track::remove_keypoint)
            if prev_window is not None:
                alive = {kf[0] for kf in r["window"]} | {sid for kf in r["window"] for sid, _nt in kf[2]}
                gone = ({kf[0] for kf in prev_window} | {sid for kf in prev_window for sid, _nt in kf[2]}) - alive
                if gone:
                    for st in state.values():
                        st[2] = [o for o in st[2] if o[0] not in gone]
            prev_window = r["window"]
            newest = r["window"][-1]
            subs = newest[2]
            if r["keyframe"] or not have_snapshot or not subs:
                continue
            if r["no_translation"] or newest[1] or any(nt for _sid, nt in subs):
                continue                                    # (the rotation-only branch builds a different problem)
            m = 0
            for sid, _nt in subs:
                assert sid in after, sid
                for _kj, tid in after[sid]:
                    st = state.get(tid)
                    if st and st[0] and st[1] and st[2] and st[2][0][1] == 1:
                        m += 1
            expected[r["frame"]] = (m, len(subs))
    by_count = {}
    for path in sorted(glob.glob(os.path.join(str(dump_dir), "*.xrba"))):
        fc = int(os.path.basename(path).split("_f")[1].split("_")[0])
        by_count.setdefault(fc, []).append(path)
    groups = [by_count[k] for k in sorted(by_count) if len(by_count[k]) == 2][-len(order):]
    assert len(groups) == len(order)
    n_checked = 0
    for (fid, _is_kf), (_first, second) in zip(order, groups):
        if fid not in expected:
            continue
        m, n_sub = expected[fid]
        d = snap.read_xrba(second)
        free = np.flatnonzero(d["frame_fix"] != 3)
        assert len(free) == n_sub and np.all(d["frame_fix"][free] == 0), second
        assert len(d["imu_i"]) == n_sub and set(d["imu_j"]) == set(free)      # keyframe -> sub 1 -> sub 2 ...: every subframe is a j once
        assert len(d["prior_frames"]) == 0 and len(d["rot_tgt"]) == 0
        assert np.all(d["landmark_fix"] == 1)
        assert np.all(np.isin(d["obs_tgt"], free)) and not np.any(np.isin(d["obs_ref"], free))
        assert len(d["obs_tgt"]) == m, "frame %d: %d reprojection prior factors over %d subframes, model predicts %d" % (
            fid, len(d["obs_tgt"]), n_sub, m)
        n_checked += 1
    assert n_checked >= 40, n_checked

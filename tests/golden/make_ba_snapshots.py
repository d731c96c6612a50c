"""Generates tests/golden/ba_snapshots/*.npz: frozen BA problems of the synthetic S1 / S2 / S3 streams (SURVEY.md 8d "S4")
with the CPU oracle's full result, asserted by tests/test_oracle_ba.py (regression pin of the oracle) and by
tests/test_ba_gpu.py (the HIP solver on exactly the problems the pipeline produces).

    python tests/golden/make_ba_snapshots.py        # needs oracle/_build (make -C oracle); no GPU

The streams run through the CPU reference pipeline (the product's host pipeline linked over the oracle,
oracle/_build/libxrslam_oracle.so) with XRSLAM_AMD_DUMP_BA set (xrslam_amd/csrc/host/ba_dump.hpp); one problem of each
shape is kept: localize_newframe (one free frame), refine_subwindow (free subframes, constant landmarks), refine_window
(free keyframes + landmarks + marginalisation prior) per stream."""
import glob
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ba_oracle as bo  # noqa: E402
from tests import ba_snapshots as snap  # noqa: E402
from xrslam_amd.harness import runner, scene  # noqa: E402
from xrslam_amd.harness.trajectory import Trajectory  # noqa: E402

CFG = os.path.join(ROOT, "configs")
ORACLE_LIB = os.path.join(ROOT, "oracle", "_build", "libxrslam_oracle.so")
STREAMS = {
    "s1": dict(n=150, slam="bench_slam_150.yaml", sensor="euroc_sensor.yaml", kw=dict(seed=1)),
    "s2": dict(n=130, slam="stress_slam_300.yaml", sensor="euroc_sensor.yaml",
               kw=dict(seed=1, traj=Trajectory(amp=1.5, speed=1.0, rot=0.8))),
    "s3": dict(n=100, slam="large_slam_600.yaml", sensor="large_sensor_1280.yaml",
               kw=dict(seed=1, w=1280, h=720, K=(780.0, 778.0, 640.0, 360.0))),
}


def classify(d):
    free_frames = int((d["frame_fix"] != 3).sum())
    free_lm = int((d["landmark_fix"] == 0).sum()) if len(d["obs_tgt"]) else 0
    if free_lm > 0 and len(d["prior_frames"]) > 0:
        return "window"
    if free_lm == 0 and free_frames == 1:
        return "localize"
    if free_lm == 0 and free_frames > 1:
        return "subwindow"
    return "other"


def main():
    out_dir = snap.SNAP_DIR
    os.makedirs(out_dir, exist_ok=True)
    for name, cfg in STREAMS.items():
        tmp = tempfile.mkdtemp(prefix="xrba_")
        os.environ["XRSLAM_AMD_DUMP_BA"] = tmp
        seq = scene.make_sequence(n_frames=cfg["n"], **cfg["kw"])
        s = runner.Session(ORACLE_LIB, seq, slam_yaml=os.path.join(CFG, cfg["slam"]), sensor_yaml=os.path.join(CFG, cfg["sensor"]))
        while s.step():
            assert not s.error(), s.error()
        s.flush()
        ate = runner.ate_rmse(list(s.poses), seq)
        s.close()
        del os.environ["XRSLAM_AMD_DUMP_BA"]
        files = sorted(glob.glob(os.path.join(tmp, "*.xrba")))
        print("%s: %d solves dumped, ATE %.4f m" % (name, len(files), ate))
        last = {}
        for f in files:   # the LAST problem of each shape: the window is full by then
            d = snap.read_xrba(f)
            last[classify(d)] = (f, d)
        want = ("localize", "subwindow", "window") if name == "s1" else ("window",)
        for kind in want:
            f, d = last[kind]
            pd = snap.to_problem(d)
            sol = pd.copy()
            sm, trace = bo.solve_trace(sol)
            arrays = {k: d[k] for k in snap.FIELDS}
            arrays.update(exp_iterations=np.int32(sm.iterations), exp_successful_steps=np.int32(sm.successful_steps),
                          exp_termination=np.int32(sm.termination), exp_initial_cost=np.float64(sm.initial_cost),
                          exp_final_cost=np.float64(sm.final_cost), exp_frame_state=sol.frame_state, exp_inv_depth=sol.inv_depth,
                          exp_trace=trace)
            path = os.path.join(out_dir, "%s_%s.npz" % (name, kind))
            np.savez_compressed(path, **arrays)
            print("  %-9s %s  F=%d L=%d M=%d MR=%d NI=%d NP=%d  -> %d iterations (%d accepted), cost %.4f -> %.4f, %d KB"
                  % (kind, os.path.basename(f), len(d["frame_state"]), len(d["inv_depth"]), len(d["obs_tgt"]), len(d["rot_tgt"]),
                     len(d["imu_i"]), len(d["prior_frames"]), sm.iterations, sm.successful_steps, sm.initial_cost, sm.final_cost,
                     os.path.getsize(path) // 1024))
        shutil.rmtree(tmp)


if __name__ == "__main__":
    main()

"""Generates tests/golden/pipeline_poses.npz: the poses the CPU reference pipeline (the product's host pipeline over the
oracle) answers on the first 64 frames of the synthetic S1 stream, in both threading modes, with the discrete counters of the
run.  tests/test_pipeline_golden.py compares the pipeline of the day with this file: an edit of the host pipeline (the order of
its hand-offs, of its id counters, of the state the feature tracker propagates) that changes a trajectory shows up there even
though every self-consistency test still passes.

    python tests/golden/make_pipeline_poses.py       # needs oracle/_build (make -C oracle); no GPU
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from xrslam_amd.harness import runner, scene  # noqa: E402

ORACLE_LIB = os.path.join(ROOT, "oracle", "_build", "libxrslam_oracle.so")
N_FRAMES = 64


def run(mode):
    seq = scene.make_sequence(n_frames=N_FRAMES, seed=1)
    s = runner.Session(ORACLE_LIB, seq, threading=mode)
    while s.step():
        assert not s.error(), s.error()
    s.flush()
    s.sync()
    t = s.times()
    counts = np.array([t.frames, t.solves, t.solve_iterations, t.marginalizations, t.keyframes], np.int64)
    poses = np.array(s.poses)
    s.close()
    return poses, counts


if __name__ == "__main__":
    pi, ci = run(0)
    pp, cp = run(1)
    out = os.path.join(ROOT, "tests", "golden", "pipeline_poses.npz")
    np.savez_compressed(out, inline_poses=pi, inline_counts=ci, pipelined_poses=pp, pipelined_counts=cp)
    print("wrote", out, pi.shape, ci, pp.shape, cp)

"""Development aid: GPU pipeline vs CPU-reference pipeline (oracle-linked) on several synthetic streams.
Prints discrete counters and the largest pose difference per seed (needs an MI355X)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xrslam_amd import _lib  # noqa: E402
from xrslam_amd.harness import runner, scene  # noqa: E402

ORACLE = os.path.join(ROOT, "oracle", "_build", "libxrslam_oracle.so")
YAML = os.path.join(ROOT, "configs", "bench_slam_150.yaml")


def run(lib, seq):
    s = runner.Session(lib, seq, slam_yaml=YAML)
    while s.step():
        pass
    s.flush()
    t = s.times()
    out = (np.array(s.poses), (t.frames, t.solves, t.solve_iterations, t.marginalizations, t.keyframes), s.error())
    s.close()
    return out


for seed in [int(a) for a in sys.argv[1:]] or [2, 7, 11]:
    seq = scene.make_sequence(n_frames=160, seed=seed)
    po, co, eo = run(ORACLE, seq)
    ph, ch, eh = run(_lib.LIB_PATH, seq)
    same = co == ch and po.shape == ph.shape
    diff = float(np.abs(po[:, 1:] - ph[:, 1:]).max()) if po.shape == ph.shape else float("nan")
    print("seed %d: counters %s %s %s  max pose diff %.3e  ATE %.4f  errors %r %r" %
          (seed, co, "==" if co == ch else "!=", ch, diff, runner.ate_rmse(list(ph), seq), eo, eh))
    assert same and diff < 1e-4
print("all seeds agree")

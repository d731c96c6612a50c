import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from xrslam_amd import _lib
from xrslam_amd.harness import runner, scene
from xrslam_amd.harness.trajectory import Trajectory
seq = scene.make_sequence(n_frames=80, seed=1, traj=Trajectory(amp=1.5, speed=0.3))
for rep in range(2):
    s = runner.Session(_lib.LIB_PATH, seq, init_frames=0)
    while s.step():
        pass
    s.flush()
    t = s.times(); r = s.init_report()
    print("attempts", r.attempts, "init scope ms total %.2f" % (1e3 * t.wall_scope[11]), "per attempt %.2f" % (1e3 * t.wall_scope[11] / max(r.attempts, 1)), "solves", t.solves, "wall_frame %.1f ms" % (1e3 * t.wall_frame))
    s.close()

"""Development aid: where does a pipelined native replay lose tens of milliseconds?  Chunks of 10 frames of the S1 bench stream through
XRSLAMAmdInstanceReplay, wall time per chunk with the counters' deltas; prints the slow chunks."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xrslam_amd.harness import runner, scene  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 640
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
seq = scene.make_sequence(n_frames=n + 1, seed=1, w=752, h=480, workers=16)
from xrslam_amd import _lib  # noqa: E402
s = runner.Session(_lib.LIB_PATH, seq, slam_yaml=os.path.join(ROOT, "configs", "bench_slam_150.yaml"), instance=True, threading=mode)
rows = []
prev = s.times()
t_start = time.perf_counter()
for k in range(0, n, 10):
    t0 = time.perf_counter()
    s.step_n(10)
    s.sync()
    dt = 1e3 * (time.perf_counter() - t0)
    t = s.times()
    rows.append((k, dt, t.keyframes - prev.keyframes, t.solve_iterations - prev.solve_iterations, t.marginalizations - prev.marginalizations,
                 1e3 * (t.wall_scope[15] - prev.wall_scope[15]), time.perf_counter() - t_start))
    prev = t
med = sorted(r[1] for r in rows)[len(rows) // 2]
print("mode", mode, "median ms per 10 frames", round(med, 2))
for r in rows:
    if r[1] > 2.5 * med:
        print("frames %d..%d: %.1f ms  keyframes %d iterations %d marginalisations %d backend_wait %.1f ms, %.3f s after the first frame" % (r[0], r[0] + 10, r[1], r[2], r[3], r[4], r[5], r[6]))
s.close()

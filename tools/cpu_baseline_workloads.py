"""CPU reference pipeline (host pipeline over the oracle, gcc -O2, our restatement) on the bench workloads S1 / S2 / S3, for
BASELINE.md's table: frames/s on one core (inline, the reference's PC build) and with the image / LK point loops on the host's
cores plus the pipelined mode.  No GPU.    python tools/cpu_baseline_workloads.py [frames]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from xrslam_amd.harness import runner, scene  # noqa: E402
from xrslam_amd.harness.scene import Trajectory  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 140
lib = os.path.join(ROOT, "oracle", "_build", "libxrslam_oracle.so")
cores = min(16, len(os.sched_getaffinity(0)))
out = {"frames_timed": n - 40, "host": open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t"), "threads": cores}
for key, wl in bench.WORKLOADS.items():
    kw = dict(w=wl["w"], h=wl["h"])
    if wl["K"]:
        kw["K"] = wl["K"]
    if wl["traj"]:
        kw["traj"] = Trajectory(**wl["traj"])
    seq = scene.make_sequence(n_frames=n + 1, seed=1, **kw)
    res = {}
    for label, threads, mode in (("1 core, inline", 1, 0), ("%d threads, pipelined" % cores, cores, 1)):
        os.environ["XR_ORACLE_THREADS"] = str(threads)
        s = runner.Session(lib, seq, slam_yaml=os.path.join(ROOT, "configs", wl["slam"]), sensor_yaml=os.path.join(ROOT, "configs", wl["sensor"]),
                           threading=mode)
        for _ in range(40):
            s.step()
        s.sync()
        t0 = time.perf_counter()
        t_w = s.times()
        for _ in range(n - 40):
            s.step()
        s.sync()
        dt = time.perf_counter() - t0
        t_e = s.times()
        it = max(1, t_e.solve_iterations - t_w.solve_iterations)
        res[label] = {"frames_per_s": round((n - 40) / dt, 2), "ms_per_ba_iteration": round(1e3 * (t_e.wall_solve - t_w.wall_solve) / it, 4),
                      "ate_mm": round(1e3 * runner.ate_rmse(s.poses, seq), 2)}
        s.close()
    out[key] = res
    print(key, res, flush=True)
print(json.dumps(out))

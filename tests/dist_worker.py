"""Worker for tests/test_dist_cpu.py: world_size-2 gloo run of the multi-sequence runner logic on CPU."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from oracle import klt_oracle as ko  # noqa: E402
from tests.util import noise_image, warp_affine  # noqa: E402
from xrslam_amd.harness.dist import RunGroup  # noqa: E402


def eleven():
    """BASELINE config 4 in miniature: 11 independent sequences round-robined over the ranks; a rank's share runs as the members of
    ONE instance group (XRSLAMAmdGroup: on a GPU they share their per-frame launches; a rank of the 8-GPU node carries one or two
    sequences, a rank of a smaller world more), every member on a thread of its own through the instance-scoped entry points --
    here with the CPU reference build of the library (same API; its group has nothing to batch); the only communication is the
    barrier pair and the metrics reduction."""
    import threading

    from xrslam_amd.harness import runner, scene
    lib = os.path.join(ROOT, "oracle", "_build", "libxrslam_oracle.so")
    g = RunGroup(backend="gloo")
    mine = g.assign(list(range(11)))
    n = 46
    seqs = {k: scene.make_sequence(n_frames=n, seed=20 + k, w=320, h=240, K=(195.0, 194.5, 160.0, 120.0)) for k in mine}
    sensor = os.path.join(ROOT, "tests", "golden", "small_sensor_320.yaml")
    g.barrier()
    t0 = time.perf_counter()
    stats = {}

    grp = runner.Group(lib)

    def work(k):
        s = runner.Session(lib, seqs[k], slam_yaml=os.path.join(ROOT, "configs", "bench_slam_150.yaml"), sensor_yaml=sensor, instance=True,
                           group=grp)
        while s.step():
            pass
        s.flush()
        err = s.error()
        P = np.array(s.poses)
        s.close()
        assert not err, err
        P = P[np.abs(P[:, 4:8]).sum(1) > 0]        # results before the first tracked frame are the all-zero pose (detail.cpp:165-168)
        idx = np.clip(np.searchsorted(seqs[k]["cam_t"], P[:, 0] - 1e-6), 0, n - 1)
        stats[k] = (len(P), float(((P[:, 1:4] - seqs[k]["states"][idx, 4:7]) ** 2).sum()))
    th = [threading.Thread(target=work, args=(k,)) for k in mine]      # the rank's whole share: one group, a thread per member
    for t in th:
        t.start()
    for t in th:
        t.join()
    grp.close()
    g.barrier()
    sec = time.perf_counter() - t0
    assert sorted(stats) == sorted(mine)
    out = g.reduce_metrics(n * len(mine), sec, sq_err_sum=sum(v[1] for v in stats.values()), n_poses=sum(v[0] for v in stats.values()))
    if g.rank == 0:
        print(json.dumps(dict(out, world=g.world, mine=mine, my_seconds=sec)))
    g.close()


def main():
    if os.environ.get("DIST_MODE") == "eleven":
        return eleven()
    g = RunGroup(backend="gloo")
    seqs = g.assign(list(range(5)))            # 5 "sequences" over 2 ranks -> 3 + 2
    g.barrier()
    t0 = time.perf_counter()
    frames = 0
    for s in seqs:                             # each rank tracks its own (tiny) sequences, no exchange
        a = noise_image(320, 240, seed=100 + s)
        b = warp_affine(a, np.eye(2), np.array([1.0, -0.5]))
        A, B = ko.OracleImage(a), ko.OracleImage(b)
        A.preprocess()
        B.preprocess()
        kp = A.detect_keypoints(np.zeros((0, 2)), 50, 20.0)
        nx, st = A.track_keypoints(B, kp, None)
        assert st.sum() > 10
        frames += 2
    g.barrier()
    sec = time.perf_counter() - t0 + 0.01 * g.rank
    out = g.reduce_metrics(frames, sec, sq_err_sum=float(g.rank + 1), n_poses=frames)
    if g.rank == 0:
        print(json.dumps(dict(out, world=g.world, mine=seqs, my_seconds=sec)))
    g.close()


if __name__ == "__main__":
    main()

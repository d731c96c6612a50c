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


def main():
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

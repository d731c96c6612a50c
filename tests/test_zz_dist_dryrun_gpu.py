"""What the 8-GPU box will run, dry, on one device: `bench.py --gpus 2 --backend gloo --sequences-per-gpu 6` --
BASELINE config 4's shape (11 sequences over the GPUs of a node, some carrying several; here 12 over 2 ranks) through the launcher,
one process per rank, every rank's sequences as the members of one instance group, the run-level barrier and the SUM / MAX reduction.
Both ranks share the box's one GPU (gloo allows that; RCCL refuses two ranks on one device) -- no scaling figure is read off this.
Asserted: rank 0 alone prints, n_gpus / per_rank / frames SUMmed over both ranks, wall time the MAX, and every one of the 12 members'
poses bit-identical to a solo run of the same stream in this process."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH_YAML = os.path.join(ROOT, "configs", "bench_slam_150.yaml")
S, STEPS, WARMUP = 6, 20, 5


@pytest.mark.gpu
def test_two_ranks_of_six_grouped_sequences_through_the_launcher(tmp_path):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    dump = str(tmp_path / "poses")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--sequences-per-gpu", str(S),
           "--steps", str(STEPS), "--warmup", str(WARMUP), "--cpu-frames", "0", "--dump-poses", dump]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                     # rank 0 alone prints
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["sequences_per_gpu"] == S
    assert [r["rank"] for r in out["per_rank"]] == [0, 1]
    # SUM of the frames over ranks / MAX of the wall time: value = 2 * S * STEPS / max(seconds)
    t_max = max(r["seconds"] for r in out["per_rank"])
    assert abs(out["value"] - 2 * S * STEPS / t_max) <= 0.02 * out["value"]
    assert abs(sum(r["frames_per_s"] * r["seconds"] for r in out["per_rank"]) - 2 * S * STEPS) < 1.0
    assert out["group"]["chain"]["requests"] >= S * STEPS      # rank 0's members went through its group
    # every member of both ranks == the same stream alone (instance, native replay, same number of frames)
    from xrslam_amd import _lib
    from xrslam_amd.harness import runner, scene
    n_frames = out["config"]["untimed_preroll_frames"] + WARMUP + STEPS
    workers = max(1, min(16, len(os.sched_getaffinity(0))))
    for rank in range(2):
        for i in range(S):
            got = np.load(os.path.join(dump, "poses_rank%d_seq%d.npy" % (rank, i)))
            seq = scene.make_sequence(n_frames=n_frames + 1, seed=1 + rank * S + i, w=752, h=480, workers=workers)
            s = runner.Session(_lib.LIB_PATH, seq, slam_yaml=BENCH_YAML, instance=True)
            s.step_n(n_frames)                       # (returns the poses it recorded: none before the seeded window is up)
            assert s.frame_k == n_frames
            s.sync()
            assert not s.error(), s.error()
            want = np.array(s.poses)
            s.close()
            assert len(want) >= STEPS + WARMUP
            np.testing.assert_array_equal(got, want, err_msg="rank %d member %d" % (rank, i))

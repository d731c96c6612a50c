"""N>1 path on CPU: world_size 2 over gloo.  One process per (would-be) GPU, sequences sharded round-robin,
no data-path collective; only a barrier and a SUM/MAX reduction of the metrics vector."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_gloo_run():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "tests", "dist_worker.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["world"] == 2
    assert out["mine"] == [0, 2, 4]                   # rank 0's round-robin share of 5 sequences
    assert out["frames"] == 10 and out["n_poses"] == 10   # SUM over ranks: (3 + 2) sequences x 2 frames
    assert out["sq_err_sum"] == 3.0
    assert out["seconds"] >= out["my_seconds"]        # MAX over ranks


def test_eleven_sequences_over_two_ranks():
    """BASELINE config 4's sharding (11 EuRoC sequences over the GPUs of a node, some carrying two): 11 synthetic sequences
    over 2 gloo ranks, every rank running its share two at a time through the instance-scoped entry points of the CPU
    reference build.  Frames and squared errors are SUMmed over the ranks, the wall time is the MAX."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29543", HSA_ENABLE_IPC_MODE_LEGACY="0", DIST_MODE="eleven")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29543", os.path.join(ROOT, "tests", "dist_worker.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["world"] == 2 and out["mine"] == [0, 2, 4, 6, 8, 10]
    assert out["frames"] == 11 * 46
    assert out["n_poses"] >= 11 * 5                       # every sequence reached TRACKING
    rmse = (out["sq_err_sum"] / out["n_poses"]) ** 0.5    # unaligned position error over all sequences
    assert rmse < 0.06
    assert out["seconds"] >= out["my_seconds"] - 1e-9

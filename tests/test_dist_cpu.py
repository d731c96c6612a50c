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


def test_bench_gpus_flag_launches_that_many_ranks():
    """`python bench.py --gpus 2` with no launcher around it must BECOME two ranks (it re-executes itself under
    torch.distributed.run, the driver's own command line) -- the flag used to be parsed and never read, so a node-level run
    would have printed n_gpus: 1.  --rendezvous-only stops after the process group / barrier / metric reduction, which is all a
    box without a GPU can run; the measuring path behind it is the same code with world > 1."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--rendezvous-only", "--steps", "7"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                 # rank 0 alone prints
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["frames"] == 14      # SUM over both ranks
    assert out["seconds_max"] == 1.25                      # MAX over ranks (rank 1 reports 1.25)
    # what each rank measured, gathered into rank 0's line (the measuring path prints the same field: a slow rank shows)
    assert [r["rank"] for r in out["per_rank"]] == [0, 1] and [r["device"] for r in out["per_rank"]] == [0, 1]
    assert out["per_rank"][0]["frames_per_s"] == 7.0 and out["per_rank"][1]["frames_per_s"] == 5.6


def test_bench_refuses_a_gpus_flag_that_contradicts_the_launcher():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"], env=env,
                       capture_output=True, text=True, timeout=60)
    assert p.returncode != 0 and "contradicts" in p.stderr

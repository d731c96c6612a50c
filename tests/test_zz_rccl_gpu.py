"""RunGroup (xrslam_amd/harness/dist.py) over the nccl (= RCCL) backend on the device: constructor with the communicator bound to the
selected GPU, barrier, SUM / MAX reduction of the metrics vector, per-rank gather -- the run-level collectives of the multi-GPU bench
(SURVEY.md section 8e), with the one rank a single-GPU box allows (RCCL refuses two ranks on one device).  Runs tools/check_rccl.py in a
process of its own: a process group is process-global state."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_run_group_collectives_over_rccl_single_rank():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29593", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_rccl.py")], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "rccl ok {'frames': 150, 'seconds': 1.25, 'sq_err_sum': 0.5, 'n_poses': 150} [[1520.0, 0.0, 2400.0]] device cuda:0" in r.stdout

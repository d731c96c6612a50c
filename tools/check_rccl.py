"""Development aid: RunGroup (xrslam_amd/harness/dist.py) over the nccl (= RCCL) backend with a single rank -- the constructor, barrier,
metric reduction and per-rank gather of the multi-GPU run, on a one-GPU box (needs an MI355X)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from xrslam_amd.harness.dist import RunGroup

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29591")
torch.cuda.set_device(0)
g = RunGroup(backend="nccl", force_init=True)
g.barrier()
m = g.reduce_metrics(150, 1.25, sq_err_sum=0.5, n_poses=150)
rows = g.gather_rows([1520.0, 0.0, 2400.0])
torch.cuda.synchronize()
assert m == dict(frames=150, seconds=1.25, sq_err_sum=0.5, n_poses=150), m
assert rows == [[1520.0, 0.0, 2400.0]], rows
print("rccl ok", m, rows, "device", g.device)
g.close()

"""Development aid: RunGroup's collectives over the nccl (= RCCL) backend with a single rank (needs an MI355X)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29591")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
s = torch.tensor([150.0, 0.5, 150.0], dtype=torch.float64, device="cuda")
m = torch.tensor([1.25], dtype=torch.float64, device="cuda")
dist.barrier()
dist.all_reduce(s, op=dist.ReduceOp.SUM)
dist.all_reduce(m, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
print("rccl ok", s.tolist(), m.tolist())
dist.destroy_process_group()

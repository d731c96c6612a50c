"""One process per GPU, one independent sequence per process.  The data path has
no exchange step; torch.distributed (backend "nccl" == RCCL over xGMI on ROCm, "gloo" on CPU) is used only
for run-level barriers and the reduction of a small metrics vector."""
import os


class RunGroup:
    def __init__(self, backend=None, force_init=False):
        """force_init: join a process group even when the world is one rank (tools/check_rccl.py: the same code path the
        multi-GPU run takes, exercised on a single-GPU box)."""
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        self.device = "cpu"
        if self.world > 1 or force_init:
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            kw = {}
            if backend == "nccl":
                # bind the communicator to the device the caller selected (torch.cuda.set_device before this constructor): no
                # guessing from the global rank, and the barrier runs on that device
                self.device = torch.device("cuda", torch.cuda.current_device())
                kw["device_id"] = self.device
            if force_init:
                kw.update(rank=self.rank, world_size=self.world)
            dist.init_process_group(backend=backend, **kw)
            self.dist = dist

    def assign(self, items):
        """Round-robin sharding of independent sequences over the ranks."""
        return [it for i, it in enumerate(items) if i % self.world == self.rank]

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def reduce_metrics(self, frames, seconds, sq_err_sum=0.0, n_poses=0):
        """frames / squared error / pose count are summed, wall seconds take the MAX over ranks."""
        if self.dist is None:
            return dict(frames=frames, seconds=seconds, sq_err_sum=sq_err_sum, n_poses=n_poses)
        import torch
        s = torch.tensor([float(frames), float(sq_err_sum), float(n_poses)], dtype=torch.float64, device=self.device)
        m = torch.tensor([float(seconds)], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(s, op=self.dist.ReduceOp.SUM)
        self.dist.all_reduce(m, op=self.dist.ReduceOp.MAX)
        return dict(frames=int(round(s[0].item())), seconds=float(m[0].item()), sq_err_sum=float(s[1].item()),
                    n_poses=int(round(s[2].item())))

    def gather_rows(self, row):
        """Every rank's `row` (a short list of floats), in rank order, on every rank: the per-rank diagnostics of a run
        (frames/s, device index, clocks) that make a slow rank visible in rank 0's line.  One all_gather of len(row) doubles."""
        if self.dist is None:
            return [list(map(float, row))]
        import torch
        mine = torch.tensor([float(v) for v in row], dtype=torch.float64, device=self.device)
        out = [torch.zeros_like(mine) for _ in range(self.world)]
        self.dist.all_gather(out, mine)
        return [[float(v) for v in t.tolist()] for t in out]

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()

"""Region sharding across the GPUs of one node.

`process_batch` regions are independent (main.cpp:458-475), so multi-GPU is plain data parallelism
over regions: no collective on the data path.  Ranks only meet for the benchmark's barrier and the
max-over-ranks clock (torch.distributed; backend "nccl" = RCCL on the GPU box, "gloo" in CPU tests).
Assignment mirrors the reference's own balance rule -- reads AND positions (main.cpp:1390-1392) --
with a greedy longest-processing-time pass over contiguous region runs.
"""
import os


def plan_shards(region_costs, world_size):
    """region_costs: list of (n_reads, n_positions).  Returns per-rank lists of region indices.

    Greedy LPT: regions sorted by cost descending go to the least-loaded rank; each rank's list is
    then sorted so its output is a sequence of genome-ordered slices (host-side concatenation restores
    the global order, like `bcftools concat -n` in uvcTN.sh:100)."""
    cost = [r * 2 + p for r, p in region_costs]   # a read costs about as much as two positions of fixed overhead
    order = sorted(range(len(cost)), key=lambda i: (-cost[i], i))
    load = [0] * world_size
    out = [[] for _ in range(world_size)]
    for i in order:
        k = min(range(world_size), key=lambda j: (load[j], j))
        out[k].append(i)
        load[k] += cost[i]
    return [sorted(v) for v in out]


def dist_env():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


class Clock:
    """barrier + max-over-ranks helper; a no-op for world_size 1."""

    def __init__(self, backend=None):
        self.rank, self.local_rank, self.world = dist_env()
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            if not dist.is_initialized():
                dist.init_process_group(backend=backend or "nccl")
            self.dist = dist
            self.backend = dist.get_backend()

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, value):
        if self.dist is None:
            return value
        import torch
        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = torch.tensor([value], dtype=torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value):
        if self.dist is None:
            return value
        import torch
        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = torch.tensor([value], dtype=torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self.dist is not None and self.dist.is_initialized():
            self.dist.destroy_process_group()

"""Region sharding across the GPUs of one node.

`process_batch` regions are independent (main.cpp:458-475), so multi-GPU is plain data parallelism
over regions: no collective on the data path.  Ranks only meet for the benchmark's barrier and the
max-over-ranks clock (torch.distributed; backend "nccl" = RCCL on the GPU box, "gloo" in CPU tests).
Assignment mirrors the reference's own balance rule -- reads AND positions (main.cpp:1390-1392): the ordered tile list is cut into
contiguous runs of about equal cost (`plan_contiguous`, the same planner as `uvc1-mi355x --shard i/n`), so that the shard outputs
concatenate in genome order like `bcftools concat -n` in uvcTN.sh:100 (`concat_bgzf`).
"""
import os


def plan_contiguous(costs, n_shards):
    """Cuts an ordered tile list into n_shards contiguous runs of about equal total cost (uvcio_plan_shards, the planner uvc1-mi355x
    --shard i/n uses): returns the shard of every tile, non-decreasing, so that the shard outputs concatenate in tile order."""
    import ctypes as C
    import numpy as np
    from . import io as uio
    c = np.ascontiguousarray(costs, dtype=np.int64)
    out = np.zeros(len(c), dtype=np.int32)
    rc = uio.dll().uvcio_plan_shards(c.ctypes.data, len(c), int(n_shards), out.ctypes.data)
    if rc != 0:
        raise ValueError(uio.dll().uvcio_last_error().decode())
    return out


def tile_costs(bam, tiles):
    """Cost of every (tid, beg, end) tile as uvc1-mi355x prices it: compressed bytes the BAI linear index attributes to it + length / 8 + 1."""
    from . import io as uio
    return [int(uio.dll().uvcio_bam_region_bytes(bam.h, tid, beg, end)) + (end - beg) // 8 + 1 for tid, beg, end in tiles]


def concat_bgzf(out_path, in_paths):
    """bcftools concat -n (uvcTN.sh:100): the shard outputs one after the other, one end-of-file marker at the end."""
    import ctypes as C
    from . import io as uio
    arr = (C.c_char_p * max(1, len(in_paths)))(*[p.encode() for p in in_paths])
    if uio.dll().uvcio_bgzf_concat(out_path.encode(), arr, len(in_paths)) != 0:
        raise IOError(uio.dll().uvcio_last_error().decode())


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

    def min_over_ranks(self, value):
        if self.dist is None:
            return value
        import torch
        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = torch.tensor([value], dtype=torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return float(t.item())

    def sum_over_ranks(self, value):
        if self.dist is None:
            return value
        import torch
        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = torch.tensor([value], dtype=torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def gather_objects(self, obj):
        """Every rank's object on rank 0 (a list there, None elsewhere): the host-side gather of the shard outputs."""
        if self.dist is None:
            return [obj]
        out = [None] * self.world if self.rank == 0 else None
        self.dist.gather_object(obj, out, dst=0)
        return out

    def close(self):
        if self.dist is not None and self.dist.is_initialized():
            self.dist.destroy_process_group()

"""Multi-GPU sharding of the codec path: one process per GPU, chunks are
independent (predictor state resets per compress() call), so rank r owns the
contiguous chunk range [r*N/W, (r+1)*N/W) and the data path has NO collective.
The only exchange is an all-gather of per-rank compressed byte counts (RCCL
over xGMI when the backend is nccl; gloo in the CPU tests) from which every
rank derives the global offsets of the container -- SURVEY.md section 8(e).
"""
from dataclasses import dataclass
from typing import List


def shard_range(nchunks: int, rank: int, world: int):
    """contiguous chunk range of `rank`: [lo, hi)"""
    lo = (nchunks * rank) // world
    hi = (nchunks * (rank + 1)) // world
    return lo, hi


@dataclass
class GlobalLayout:
    rank_bytes: List[int]      # compressed bytes held by each rank
    rank_base: List[int]       # exclusive scan: global byte offset of each rank's first chunk
    total_bytes: int

    def global_offsets(self, rank, local_offsets):
        """local chunk offsets (tensor/array/list) -> offsets in the global container"""
        return local_offsets + self.rank_base[rank]


def gather_layout(local_bytes: int, device=None) -> GlobalLayout:
    """all-gather one int64 per rank (64 bytes on 8 GPUs: latency-bound, ring vs
    direct is irrelevant at this size) and exclusive-scan it."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return GlobalLayout([int(local_bytes)], [0], int(local_bytes))
    world = dist.get_world_size()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    mine = torch.tensor([int(local_bytes)], dtype=torch.int64, device=device)
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    sizes = [int(t.item()) for t in gathered]
    base, acc = [], 0
    for s in sizes:
        base.append(acc)
        acc += s
    return GlobalLayout(sizes, base, acc)


def max_over_ranks(x: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(x)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(x: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(x)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())

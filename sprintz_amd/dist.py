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


class LayoutGather:
    """The write path's one exchange, kept ready across batches: every rank contributes its container's byte
    count (a device word -- `offsets[nchunks:]` as sprintz_mi355x_compact leaves it) and receives everybody's.

    backend "rccl-c-abi": sprintz_mi355x_gather_layout = ncclAllGather over xGMI, enqueued on the compute stream
    right behind the compaction pass (the communicator is bootstrapped by shipping RCCL's 128-byte unique id
    through the process group that torchrun set up -- bootstrap only).  If RCCL cannot be brought up that way
    the same gather goes through torch.distributed (backend "torch.distributed/nccl" = RCCL as well, or gloo
    on CPU tensors in the tests); with one rank there is nothing to exchange ("single-rank")."""

    def __init__(self, device=None, prefer_c_abi=True):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size() if self.on else 1
        self.rank = dist.get_rank() if self.on else 0
        cuda = device is not None and torch.device(device).type == "cuda"
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self.all = torch.zeros(self.world, dtype=torch.int64, device=self.device)
        self.comm = None
        self.backend = "single-rank" if self.world == 1 else f"torch.distributed/{dist.get_backend()}"
        self.ranks_seen = 1
        self.c_abi_error = None
        # prefer_c_abi == "force": try the library's communicator whatever carries the bootstrap (bench.py --dry-launch --rccl on a
        # one-device box: the process group is gloo, RCCL then refuses the second rank on the device and the fall-back shows)
        if cuda and prefer_c_abi and (self.world == 1 or dist.get_backend() == "nccl" or prefer_c_abi == "force"):
            self._init_c_abi()

    def _init_c_abi(self):
        import ctypes as C
        from . import _lib
        torch, dist = self.torch, self.dist
        idbuf = (C.c_uint8 * _lib.COMM_ID_BYTES)()
        ok = 1
        if self.rank == 0:
            ok = 1 if _lib.comm_unique_id(idbuf) == 0 else 0
        boot = self.device if (self.world == 1 or dist.get_backend() == "nccl") else torch.device("cpu")   # what the bootstrap's collectives travel on
        t = torch.zeros(_lib.COMM_ID_BYTES + 1, dtype=torch.uint8, device=boot)
        if self.rank == 0:
            t[:-1] = torch.frombuffer(bytearray(idbuf), dtype=torch.uint8).to(boot)
            t[-1] = ok
        if self.world > 1:
            dist.broadcast(t, src=0)
        host = t.cpu()
        if int(host[-1]) != 1:
            self.c_abi_error = _lib.last_error() or "rank 0 could not create an RCCL unique id"
            return
        idb = (C.c_uint8 * _lib.COMM_ID_BYTES)(*host[:-1].tolist())
        comm = C.c_void_p()
        with torch.cuda.device(self.device):
            rc = _lib.comm_init(idb, self.rank, self.world, C.byref(comm))
        good = torch.tensor([1 if rc == 0 else 0], dtype=torch.int64, device=boot)
        if self.world > 1:
            dist.all_reduce(good, op=dist.ReduceOp.MIN)            # all ranks take the same path
        if int(good.item()) == 1:
            self.comm = comm
            self.backend = "rccl-c-abi (sprintz_mi355x_gather_layout: ncclAllGather, in-stream)"
        else:
            self.c_abi_error = _lib.last_error()
            if rc == 0:
                _lib.comm_destroy(comm)

    def gather_async(self, d_total):
        """d_total: 1-element int64 tensor holding this rank's byte count (a view is fine).  Enqueues the
        all-gather on the current stream (device tensors) / runs it (CPU tensors)."""
        import ctypes as C
        if self.comm is not None:
            from . import _lib
            st = C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)
            _lib.check(_lib.gather_layout(self.comm, d_total.data_ptr(), self.all.data_ptr(), st))
        elif self.world > 1:
            parts = [self.all[r:r + 1] for r in range(self.world)]          # views: the gather lands in self.all
            self.dist.all_gather(parts, d_total.reshape(1).contiguous())
        else:
            self.all[:1] = d_total.reshape(1)

    def layout(self, local_bytes=None) -> GlobalLayout:
        """the gathered counts as a GlobalLayout (synchronises).  With local_bytes given and no gather issued yet,
        performs one."""
        if local_bytes is not None:
            self.gather_async(self.torch.tensor([int(local_bytes)], dtype=self.torch.int64, device=self.device))
        sizes = [int(v) for v in self.all.cpu().tolist()]
        self.ranks_seen = len(sizes)
        base, acc = [], 0
        for s in sizes:
            base.append(acc)
            acc += s
        lay = GlobalLayout(sizes, base, acc)
        lay.bases = base
        return lay

    def close(self):
        if self.comm is not None:
            from . import _lib
            _lib.comm_destroy(self.comm)
            self.comm = None


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

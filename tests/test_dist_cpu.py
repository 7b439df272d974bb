"""CPU tests of the N>1 path (gloo, world_size 2): chunk sharding and the one
collective the codec path has -- the all-gather of per-rank compressed byte
counts that yields the global container layout.  The per-rank
compressed payload here comes from the oracle (no GPU in this tier); on the GPU
the same code runs with backend nccl (= RCCL over xGMI) in bench.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from harness import Oracle, gen_walk


def test_shard_range_partitions_exactly():
    from sprintz_amd.dist import shard_range
    for n in (0, 1, 7, 8, 10000, 131072, 6554):
        for w in (1, 2, 3, 4, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b and c <= d
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sprintz_amd.dist import LayoutGather, gather_layout, max_over_ranks, shard_range, sum_over_ranks
        codec, esz, ndims, chunk_len, nchunks = "xff", 2, 8, 5120, 40
        data = gen_walk(np.random.default_rng(42), nchunks * chunk_len, ndims, esz, 8, flat_every=4)   # same on every rank
        lo, hi = shard_range(nchunks, rank, world)
        orc = Oracle()
        mine = orc.compress_chunks(codec, data[lo * chunk_len:hi * chunk_len], chunk_len, ndims)
        local_sizes = np.array([s.size for s in mine], np.int64)
        local_offsets = np.concatenate([[0], np.cumsum(local_sizes)])[:-1]
        layout = gather_layout(int(local_sizes.sum()))
        # the object bench.py keeps across batches gives the same answer (CPU tensors: torch.distributed path)
        lg = LayoutGather(None)
        lay2 = lg.layout(int(local_sizes.sum()))
        assert lg.backend == "torch.distributed/gloo" and lg.ranks_seen == world
        assert (lay2.rank_bytes, lay2.rank_base, lay2.total_bytes) == (layout.rank_bytes, layout.rank_base, layout.total_bytes)
        lg.close()
        goffs = layout.global_offsets(rank, local_offsets)
        assert max_over_ranks(float(rank)) == world - 1
        assert sum_over_ranks(1.0) == world
        q.put((rank, layout.rank_bytes, layout.rank_base, layout.total_bytes, goffs.tolist(),
               b"".join(s.tobytes() for s in mine)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(240)
@pytest.mark.parametrize("world", [2, 4])
def test_gather_layout_gloo(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=150) for _ in range(world)])
    [p.join(30) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    # every rank derived the same layout
    assert all(r[1:4] == res[0][1:4] for r in res)
    rank_bytes, rank_base, total = res[0][1:4]
    assert rank_base == [sum(rank_bytes[:r]) for r in range(world)] and total == sum(rank_bytes)
    # the global container assembled from per-rank pieces == one-process compression of everything
    codec, esz, ndims, chunk_len, nchunks = "xff", 2, 8, 5120, 40
    data = gen_walk(np.random.default_rng(42), nchunks * chunk_len, ndims, esz, 8, flat_every=4)
    orc = Oracle()
    whole = orc.compress_chunks(codec, data, chunk_len, ndims)
    assert b"".join(r[5] for r in res) == b"".join(s.tobytes() for s in whole)
    goffs = [g for r in res for g in r[4]]
    want = np.concatenate([[0], np.cumsum([s.size for s in whole])])[:-1]
    assert goffs == want.tolist()
    # and decodes back, chunk by chunk, from the global offsets
    comp = np.frombuffer(b"".join(r[5] for r in res) + bytes(64), np.uint8)
    dec = orc.decompress_chunks(codec, comp, np.array(goffs, np.uint64), esz, chunk_len, data.size)
    assert np.array_equal(dec, data)

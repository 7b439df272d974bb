"""GPU tests (-m gpu) on the data bench.py itself times (tools/synth.py: SURVEY.md 8d's generator, splitmix64 seeded per
chunk): every BASELINE configuration at its bench size -- GPU round trip, and the compressed bytes of a strided sample of
chunks against the oracle (round 2 checked the full sizes on torch.randint walks only, and the bench never compares stream
bytes on its own data).  Also: BASELINE config 4 at 800 000 chunks (sampled segments against the writer's specification
and the oracle's Huff0 reader), config 4 on blocks written by the host's libzstd (a tree in every block) at 80 000 chunks,
and the RCCL layout gather of comm.cpp on two real ranks whenever two GPUs are visible.
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sz():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import sprintz_amd
    return sprintz_amd


def bench_input(name, device):
    """exactly what bench.py's run_config(name) / headline generates on rank 0 of 1"""
    import torch
    from synth import synth_torch
    if name == "cfg2":
        return ("xff", 2, 8, 5120, 131072), synth_torch("walk", 2, 131072, 640, 8, device, seed=123, step=8, chunk0=0)
    if name.startswith("cfg2_"):               # bench.py's data_sweep: the headline shape on SURVEY 8d's other generators
        kind, step = {"cfg2_uniform": ("uniform", 0), "cfg2_walk300": ("walk", 300), "cfg2_walkflat": ("walkflat", 8)}[name]
        return ("xff", 2, 8, 5120, 131072), synth_torch(kind, 2, 131072, 640, 8, device, seed=123, step=step, chunk0=0)
    if name == "cfg1":
        return ("delta", 1, 1, 1024, 524288), synth_torch("walk", 1, 524288, 1024, 1, device, seed=123, step=2, chunk0=0)
    if name == "cfg3_10k":
        return ("delta", 1, 80, 10240, 52429), synth_torch("walk", 1, 52429, 128, 80, device, seed=123, step=2, chunk0=0)
    if name == "cfg3_1k":                      # 1024 elements do not hold whole rows of 80: 64-row series, cut every 1024
        from synth import synth_cut_rows
        return ("delta", 1, 80, 1024, 524288), synth_cut_rows("walk", 1, 524288, 1024, 80, device, seed=123, step=2)
    raise ValueError(name)


@pytest.mark.parametrize("name", ["cfg2", "cfg2_uniform", "cfg2_walk300", "cfg2_walkflat", "cfg1", "cfg3_1k", "cfg3_10k"])
def test_bench_inputs_roundtrip_and_sample_parity(sz, oracle, name):
    import torch
    (codec, esz, ndims, chunk_len, nchunks), x = bench_input(name, "cuda:0")
    cd = sz.ChunkedCodec(codec, esz, ndims, chunk_len, device="cuda:0")
    batch = cd.compress(x)
    rets = torch.empty(nchunks, dtype=torch.int64, device="cuda:0")
    out = cd.decompress(batch, rets=rets)
    assert torch.equal(out.view(torch.uint8), x.view(torch.uint8)), name
    assert bool((rets == chunk_len).all().item())
    sizes, offs = batch.sizes.cpu().numpy(), batch.offsets.cpu().numpy()
    sample = np.unique(np.concatenate([np.arange(0, nchunks, max(1, nchunks // 200)), [nchunks - 1]]))
    xs = x.view(torch.uint8).reshape(nchunks, chunk_len * esz)[torch.from_numpy(sample).cuda()].cpu().numpy()
    comp = batch.data.cpu().numpy()
    dt = np.uint8 if esz == 1 else np.uint16
    for j, c in enumerate(sample):
        want, _ = oracle.compress(codec, xs[j].view(dt), ndims)
        assert sizes[c] == want.size, (name, c)
        assert np.array_equal(comp[offs[c]:offs[c] + sizes[c]], want), (name, c)
    ratio = x.numel() * esz / float(sizes.astype(np.int64).sum())
    # the ratios bench.py prints for these inputs (BASELINE.md / DESIGN.md 5): a generator that drifted would show here
    lo, hi = {"cfg2": (2.7, 3.0), "cfg1": (2.2, 2.5), "cfg3_1k": (0.98, 1.0), "cfg3_10k": (2.1, 2.4),
              "cfg2_uniform": (0.96, 0.98), "cfg2_walk300": (1.4, 1.6), "cfg2_walkflat": (3.0, 6.0)}[name]
    assert lo < ratio < hi, (name, ratio)


def test_bench_input_cfg5_colmajor(sz, oracle):
    """1 M rows x 32 variables stored column by column, 160-row chunks, as bench_cfg5 builds it"""
    import torch
    from synth import synth_torch
    D, rpc, nrows = 32, 160, 1 << 20
    n = (nrows + rpc - 1) // rpc
    full = synth_torch("walk", 2, n, rpc, D, "cuda:0", seed=123, step=8, chunk0=0).view(n, rpc, D)
    cols = full.permute(2, 0, 1).reshape(D, n * rpc)[:, :nrows].contiguous()
    cd = sz.ChunkedCodec("xff", 2, D, rpc * D, device="cuda:0")
    batch = cd.compress_colmajor(cols)
    back = cd.decompress_colmajor(batch)
    assert torch.equal(back.view(torch.int16), cols.view(torch.int16))
    sizes, offs, comp = batch.sizes.cpu().numpy(), batch.offsets.cpu().numpy(), batch.data.cpu().numpy()
    rows_h = full.reshape(n * rpc, D)[:nrows].cpu().numpy()               # the row-major flattening the reference would see
    for c in list(range(0, n, 97)) + [n - 1]:
        r0, r1 = c * rpc, min((c + 1) * rpc, nrows)
        want, _ = oracle.compress("xff", rows_h[r0:r1].reshape(-1), D)
        assert sizes[c] == want.size, c
        assert np.array_equal(comp[offs[c]:offs[c] + sizes[c]], want), c


def test_cfg4_chain_at_800000_chunks(sz, oracle):
    """BASELINE config 4 at the largest batch bench.py runs: 800 000 chunks (8.2 GB of samples) through samples -> Sprintz
    streams -> Huff0 blocks -> streams -> samples on the GPU; whole 64-chunk segments sampled across the batch against the
    writer's specification, chunks of them through the oracle's Huff0 reader."""
    import torch
    from synth import synth_torch
    nchunks, chunk_len, ndims = 800000, 5120, 8
    x = synth_torch("walk", 2, nchunks, chunk_len // ndims, ndims, "cuda:0", seed=123, step=8, chunk0=0)
    cd = sz.ChunkedCodec("xff", 2, ndims, chunk_len, device="cuda:0")
    batch = cd.compress(x)
    blocks, bo = sz.huf0_compress(batch)
    sizes = batch.sizes.to(torch.int64)
    oo = torch.zeros(nchunks + 1, dtype=torch.int64, device="cuda")
    oo[1:] = torch.cumsum(sizes, 0)
    rets = torch.empty(nchunks, dtype=torch.int64, device="cuda")
    streams = sz.huf0_decompress(blocks, bo, oo, rets=rets)
    assert torch.equal(rets, sizes)
    out = torch.empty(nchunks * chunk_len, dtype=torch.uint16, device="cuda")
    cd.decompress_into(streams, oo, nchunks, out)
    assert torch.equal(out.view(torch.int16), x.view(torch.int16))
    del out, x
    assert int(bo[-1].item()) < int(oo[-1].item())
    nseg = nchunks // 64
    bo_h, offs_h, sz_h = bo.cpu().numpy(), batch.offsets.cpu().numpy().astype(np.uint64), batch.sizes.cpu().numpy()
    for seg in [0, 1, nseg // 7, nseg // 3, nseg // 2, nseg - 2, nseg - 1]:
        c0 = 64 * seg
        lo, hi = int(offs_h[c0]), int(offs_h[c0 + 64])
        comp = batch.data[lo:hi + 64].cpu().numpy()
        want, wo = oracle.huf0_compress(comp, offs_h[c0:c0 + 65] - np.uint64(lo), sz_h[c0:c0 + 64])
        got = blocks[int(bo_h[c0]):int(bo_h[c0 + 64])].cpu().numpy()
        assert np.array_equal(got, want), seg
        for c in (c0, c0 + 31, c0 + 63):
            b = got[int(bo_h[c] - bo_h[c0]):int(bo_h[c + 1] - bo_h[c0])]
            plain, ret = oracle.huf0_decompress(b, int(sz_h[c]))
            o = int(offs_h[c]) - lo
            assert ret == sz_h[c] and np.array_equal(plain, comp[o:o + int(sz_h[c])]), c


@pytest.mark.parametrize("nt", [80000, 800000])
def test_cfg4_chain_on_libzstd_blocks(sz, nt):
    """the blocks lzbench's huff0 would hand over: HUF_compress of the host's libzstd, one call per chunk, a tree of its own in
    every block (the writer of this library repeats a tree per 64-chunk segment, which its reader exploits).  2 048 distinct
    chunks' streams coded on the host, tiled to 80 000 / 800 000 blocks (the largest batch bench.py runs), decoded on the GPU to
    streams and on to samples."""
    import torch
    from synth import synth_torch
    try:
        z = C.CDLL("libzstd.so.1")
        z.HUF_compress.restype = C.c_size_t
        z.HUF_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        z.HUF_isError.restype = C.c_uint
        z.HUF_isError.argtypes = [C.c_size_t]
    except (OSError, AttributeError):
        pytest.skip("no libzstd.so.1 exporting HUF_compress on this host")
    nd, chunk_len, ndims = 2048, 5120, 8
    x = synth_torch("walk", 2, nd, chunk_len // ndims, ndims, "cuda:0", seed=123, step=8, chunk0=0)
    cd = sz.ChunkedCodec("xff", 2, ndims, chunk_len, device="cuda:0")
    batch = cd.compress(x)
    comp, offs, sz_h = batch.data.cpu().numpy(), batch.offsets.cpu().numpy().astype(np.int64), batch.sizes.cpu().numpy().astype(np.int64)
    blocks, tmp, coded = [], np.zeros(1 << 16, np.uint8), 0
    for c in range(nd):
        st = np.ascontiguousarray(comp[offs[c]:offs[c] + sz_h[c]])
        r = z.HUF_compress(tmp.ctypes.data, tmp.size, st.ctypes.data, st.size)
        if r == 0 or z.HUF_isError(r):
            blocks.append(st)                                              # not compressible: stored
        else:
            blocks.append(tmp[:r].copy())
            coded += 1
    assert coded > nd * 0.9
    reps = (nt + nd - 1) // nd
    order = np.tile(np.arange(nd), reps)[:nt]
    bsz = np.array([b.size for b in blocks], np.int64)[order]
    bo = np.zeros(nt + 1, np.int64)
    bo[1:] = np.cumsum(bsz)
    oo = np.zeros(nt + 1, np.int64)
    oo[1:] = np.cumsum(sz_h[order])
    one = torch.from_numpy(np.concatenate(blocks)).cuda()
    plain_one = torch.from_numpy(np.concatenate([comp[offs[c]:offs[c] + sz_h[c]] for c in range(nd)])).cuda()
    d_blocks = torch.cat([one.repeat(reps)[: int(bo[-1])], torch.zeros(64, dtype=torch.uint8, device="cuda")])
    rets = torch.empty(nt, dtype=torch.int64, device="cuda")
    oo_d = torch.from_numpy(oo).cuda()
    streams = sz.huf0_decompress(d_blocks, torch.from_numpy(bo).cuda(), oo_d, rets=rets)
    assert torch.equal(rets.cpu(), torch.from_numpy(sz_h[order]))
    assert torch.equal(streams[: int(oo[-1])], plain_one.repeat(reps)[: int(oo[-1])])
    out = torch.empty(nt * chunk_len, dtype=torch.uint16, device="cuda")
    cd.decompress_into(streams, oo_d, nt, out)
    want = x.view(torch.int16).reshape(nd, chunk_len)
    got = out.view(torch.int16).reshape(nt, chunk_len)
    for r in range(reps):
        lo, hi = r * nd, min((r + 1) * nd, nt)
        assert torch.equal(got[lo:hi], want[: hi - lo]), r


def _two_rank_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", device_id=dev)
    try:
        from sprintz_amd.dist import LayoutGather
        g = LayoutGather(dev)
        mine = torch.tensor([1000 + 17 * rank], dtype=torch.int64, device=dev)
        g.gather_async(mine)
        lay = g.layout()
        q.put((rank, g.backend, lay.rank_bytes, lay.rank_base, g.c_abi_error))
        g.close()
    finally:
        dist.destroy_process_group()


def test_layout_gather_on_two_real_ranks(sz):
    """comm.cpp's ncclAllGather with a communicator of more than one rank (round 2 only ever saw one): needs two GPUs"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible: the two-rank RCCL gather needs two")
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    got = sorted(q.get(timeout=300) for _ in range(2))
    [p.join(60) for p in procs]
    for rank, backend, sizes, bases, err in got:
        assert backend.startswith("rccl-c-abi"), (backend, err)
        assert sizes == [1000, 1017] and bases == [0, 1000]


def test_the_line_of_an_8_rank_run_on_one_device():
    """`python bench.py --gpus 8` as the driver starts it at round end, on the ONE GPU a test box has (BENCH_ONE_DEVICE: eight ranks
    share the device, gloo carries the layout exchange -- RCCL refuses two ranks on one device): the sharding, the strong-scaled
    per_config legs, the merge over ranks and the record are the real ones.  The stdout line must stay below the driver's parser
    limit WITH eight ranks' fields on it, name all eight container bases, and say who gathered the layout."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"BENCH_ONE_DEVICE": "1", "BENCH_BACKEND": "gloo"})
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--nchunks", "8192", "--steps", "5", "--warmup", "2",
                        "--configs", "cfg4_10000,cfg5", "--no-sweep", "--no-extras", "--cpu-seconds", "0.5", "--config-reps", "5", "--ramp-ms", "20"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    assert len(lines[0]) < 10000, len(lines[0])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["config"]["chunks_per_gpu"] == 8192
    assert len(d["rank_bases"]) == 8 and d["rank_bases"][0] == 0 and all(b > a for a, b in zip(d["rank_bases"], d["rank_bases"][1:]))
    assert "rccl_ranks_seen" in d and d["container_bytes_all_ranks"] > d["rank_bases"][7]
    assert d["roofline"]["traffic_measured_in_this_run"] is False
    summ = d["per_config_summary"]
    assert set(summ) >= {"fields", "cfg2", "cfg4_10000", "cfg5"} and summ["cfg4_10000"] != "error" and summ["cfg5"] != "error"
    assert d["value"] > 0 and d["roofline"]["frac"] > 0

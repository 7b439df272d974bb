"""GPU tests (-m gpu): genuine Huff0 blocks (written by libzstd's HUF_compress; committed under
tests/golden/) through sprintz_mi355x_huf0_decompress_batch, against the plain bytes and the
oracle, then on into the Sprintz decoder.  Nothing here needs libzstd or /root/reference."""
import numpy as np
import pytest

from harness import DTYPES

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["sync form", "sync form, hinted", "single-wave form", "big-batch form"])
def sz(request):
    """every test on every stream kernel: the one-wave-per-chunk form with self-synchronising decoders (huf0_sync.h: what batches up to
    8 192 chunks get; without a size hint its blocks above 4 KB are read from global memory, with one they sit in LDS), the single-wave form
    it replaced there (a wave per 16 chunks), and the bandwidth-sized form forced (2-wave workgroups around one table, 64-byte pieces)"""
    import functools
    import types
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import sprintz_amd
    from sprintz_amd import _lib
    assert _lib.set_option(_lib.OPT_HUF0_BIG_BATCH, 0 if request.param == "big-batch form" else 16385) == 0
    assert _lib.set_option(_lib.OPT_HUF0_SYNC_CHUNKS, (1 << 30) if request.param.startswith("sync") else 0) == 0
    mod = sprintz_amd
    if request.param == "sync form, hinted":                  # the same module with the hint filled in
        mod = types.SimpleNamespace(**{k: getattr(sprintz_amd, k) for k in dir(sprintz_amd) if not k.startswith("__")})
        mod.huf0_decompress = functools.partial(sprintz_amd.huf0_decompress, max_block_bytes=16384)
    yield mod
    # back to what the process was configured with (the environment's value, else the library's default), not to a constant
    import os
    _lib.set_option(_lib.OPT_HUF0_BIG_BATCH, 16385)            # (no environment knob: the library's constant)
    _lib.set_option(_lib.OPT_HUF0_SYNC_CHUNKS, int(os.environ.get("SPRINTZ_MI355X_HUF0_SYNC_CHUNKS", 8192)))


def pack(blocks, plains, align=1):
    import torch
    bo = np.zeros(len(blocks) + 1, np.int64)
    bo[1:] = np.cumsum([b.size for b in blocks])
    oo = np.zeros(len(plains) + 1, np.int64)
    for i, p in enumerate(plains):
        oo[i + 1] = oo[i] + p.size
    data = np.concatenate(list(blocks) + [np.zeros(16, np.uint8)])
    return torch.from_numpy(data).cuda(), torch.from_numpy(bo).cuda(), torch.from_numpy(oo).cuda(), oo


def test_committed_blocks_decode(sz, golden_huf0):
    import torch
    manifest, arrays = golden_huf0
    blocks = [arrays["b%04d" % m["idx"]] for m in manifest]
    plains = [arrays["p%04d" % m["idx"]] for m in manifest]
    d, bo, oo, oo_h = pack(blocks, plains)
    rets = torch.full((len(blocks),), -99, dtype=torch.int64, device="cuda")
    out = sz.huf0_decompress(d, bo, oo, rets=rets).cpu().numpy()
    r = rets.cpu().numpy()
    for i, m in enumerate(manifest):
        assert r[i] == plains[i].size, (m, r[i])
        assert np.array_equal(out[oo_h[i]:oo_h[i + 1]], plains[i]), m
    assert not out[oo_h[-1]:].any()


def test_damaged_blocks_are_contained(sz, oracle, golden_huf0):
    """a damaged block decodes like the oracle says or is rejected; its neighbours are untouched"""
    import torch
    manifest, arrays = golden_huf0
    rng = np.random.default_rng(12)
    coded = [m for m in manifest if m["kind"] in ("fse", "nibbles") and m["n"] <= 5000][:80]
    blocks, plains, want = [], [], []
    for k, m in enumerate(coded):
        plain, blk = arrays["p%04d" % m["idx"]], arrays["b%04d" % m["idx"]].copy()
        if k % 2:
            if k % 4 == 1:
                blk[rng.integers(0, min(blk.size, 60))] ^= 1 << rng.integers(0, 8)
            else:
                blk = blk[: rng.integers(2, blk.size)]
        got, ret = oracle.huf0_decompress(blk, plain.size)
        blocks.append(blk)
        plains.append(plain)
        want.append((got, ret))
    d, bo, oo, oo_h = pack(blocks, plains)
    rets = torch.full((len(blocks),), -99, dtype=torch.int64, device="cuda")
    out = sz.huf0_decompress(d, bo, oo, rets=rets).cpu().numpy()
    r = rets.cpu().numpy()
    rejected = 0
    for k in range(len(blocks)):
        got, ret = want[k]
        if ret < 0:
            assert r[k] < 0, (k, r[k])
            rejected += 1
        else:
            assert r[k] == plains[k].size and np.array_equal(out[oo_h[k]:oo_h[k + 1]], got), k
    assert rejected > 10


def test_every_block_as_a_segment_leader_and_as_a_follower(sz, oracle, golden_huf0):
    """The reader parses the first block of every 64-block segment with its own kernel (a wave per leader), lets blocks
    that repeat the leader's tree copy its descriptor and decodes such segments through one shared table.  Here every
    committed block -- and a damaged twin of every second one -- leads a segment of copies of itself (the follower /
    one-table path), followed by a segment it leads with strangers behind it (the per-chunk path)."""
    import torch
    manifest, arrays = golden_huf0
    rng = np.random.default_rng(5)
    coded = [m for m in manifest if m["n"] <= 6000]
    stranger = [(arrays["b%04d" % m["idx"]], arrays["p%04d" % m["idx"]]) for m in coded[:63]]
    blocks, plains, want = [], [], []
    for k, m in enumerate(coded):
        plain, blk = arrays["p%04d" % m["idx"]], arrays["b%04d" % m["idx"]].copy()
        if k % 2 and blk.size > 4 and blk.size < plain.size:
            if k % 4 == 1:
                blk[rng.integers(0, min(blk.size, 60))] ^= 1 << rng.integers(0, 8)
            else:
                blk = blk[: rng.integers(2, blk.size)]
        got, ret = oracle.huf0_decompress(blk, plain.size)
        copies = 64 if k % 3 else 17 + k % 40                 # a full segment of copies, or a short run of them ...
        for _ in range(copies):
            blocks.append(blk); plains.append(plain); want.append((got, ret))
        while len(blocks) % 64:                                # ... filled up with strangers (their own trees)
            b, p = stranger[len(blocks) % 63]
            blocks.append(b); plains.append(p); want.append((p, p.size))
    d, bo, oo, oo_h = pack(blocks, plains)
    rets = torch.full((len(blocks),), -99, dtype=torch.int64, device="cuda")
    out = sz.huf0_decompress(d, bo, oo, rets=rets).cpu().numpy()
    r = rets.cpu().numpy()
    rejected = 0
    for k in range(len(blocks)):
        got, ret = want[k]
        if ret < 0:
            assert r[k] < 0, (k, r[k])
            rejected += 1
        else:
            assert r[k] == plains[k].size, (k, r[k], plains[k].size)
            assert np.array_equal(out[oo_h[k]:oo_h[k + 1]], got), k
    assert rejected > 100


def test_huff0_then_sprintz(sz, oracle, golden_huf0):
    """the chain the paper describes: Huff0 blocks of Sprintz streams -> streams -> samples"""
    import torch
    manifest, arrays = golden_huf0
    ms = [m for m in manifest if m["name"] == "sprintz_xff_16_8"]
    assert len(ms) >= 20
    blocks = [arrays["b%04d" % m["idx"]] for m in ms]
    streams = [arrays["p%04d" % m["idx"]] for m in ms]
    d, bo, oo, oo_h = pack(blocks, streams)
    comp = sz.huf0_decompress(d, bo, oo)
    cd = sz.ChunkedCodec("xff", 2, 8, 5120, device="cuda:0")
    out = torch.empty(len(ms) * 5120, dtype=torch.uint16, device="cuda")
    cd.decompress_into(comp, oo, len(ms), out)
    want = np.concatenate([oracle.decompress("xff", s, 2, 5120)[0] for s in streams])
    assert np.array_equal(out.cpu().numpy().view(np.uint16), want.view(np.uint16))


def test_many_chunks(sz, golden_huf0):
    """a batch that spans many workgroups, odd count, empty chunks in between"""
    import torch
    manifest, arrays = golden_huf0
    base = [(arrays["b%04d" % m["idx"]], arrays["p%04d" % m["idx"]]) for m in manifest if m["n"] <= 11000]
    blocks, plains = [], []
    for k in range(5003):
        if k % 97 == 5:
            blocks.append(np.zeros(0, np.uint8)); plains.append(np.zeros(0, np.uint8))
        else:
            b, p = base[(k * 7) % len(base)]
            blocks.append(b); plains.append(p)
    d, bo, oo, oo_h = pack(blocks, plains)
    rets = torch.full((len(blocks),), -99, dtype=torch.int64, device="cuda")
    out = sz.huf0_decompress(d, bo, oo, rets=rets)
    want = torch.from_numpy(np.concatenate(plains)).cuda()
    assert torch.equal(out[: want.numel()], want)
    assert torch.equal(rets.cpu(), torch.tensor([p.size for p in plains], dtype=torch.int64))


@pytest.mark.parametrize("codec,esz,ndims,nchunks", [("xff", 2, 8, 333), ("delta", 1, 1, 200), ("delta", 1, 80, 130), ("xff", 2, 2, 65)])
def test_writer_matches_its_specification(sz, oracle, codec, esz, ndims, nchunks):
    """huf0_compress_batch writes oracle_huf0_compress_batch's bytes; the blocks decode (GPU and oracle) to the streams"""
    import torch
    from harness import gen_walk, gen_fuzz
    rng = np.random.default_rng(nchunks)
    chunk_len = 5120
    data = np.concatenate([gen_walk(rng, (nchunks - 20) * chunk_len, ndims, esz, 8, flat_every=4), gen_fuzz(rng, 10 * chunk_len, esz, 0),
                           np.zeros(10 * chunk_len - 7, DTYPES[esz])])
    cd = sz.ChunkedCodec(codec, esz, ndims, chunk_len, device="cuda:0")
    batch = cd.compress(torch.from_numpy(data).cuda())
    blocks, bo = sz.huf0_compress(batch)
    want, wo = oracle.huf0_compress(batch.data.cpu().numpy(), batch.offsets.cpu().numpy().astype(np.uint64), batch.sizes.cpu().numpy())
    bo_h = bo.cpu().numpy()
    assert np.array_equal(bo_h.astype(np.uint64), wo)
    assert np.array_equal(blocks[: int(bo_h[-1])].cpu().numpy(), want)
    # back through the GPU decoder into a byte-dense container, then to samples
    sizes = batch.sizes.cpu().numpy().astype(np.int64)
    oo = np.zeros(nchunks + 1, np.int64)
    oo[1:] = np.cumsum(sizes)
    oo_d = torch.from_numpy(oo).cuda()
    rets = torch.empty(nchunks, dtype=torch.int64, device="cuda")
    streams = sz.huf0_decompress(blocks, bo, oo_d, rets=rets)
    assert np.array_equal(rets.cpu().numpy(), sizes)
    out = torch.empty(nchunks * chunk_len, dtype=torch.uint8 if esz == 1 else torch.uint16, device="cuda")
    cd.decompress_into(streams, oo_d, nchunks, out)
    assert np.array_equal(out.cpu().numpy().view(DTYPES[esz])[: data.size], data)


@pytest.mark.parametrize("nchunks", [10000, 80000])
def test_full_size_cfg4_chain(sz, oracle, nchunks):
    """BASELINE config 4 at the batch sizes bench.py runs (10 000 chunks as stated, 80 000): samples -> Sprintz streams -> Huff0
    blocks -> streams -> samples on the GPU; on a strided sample of chunks the blocks are what the writer's specification says
    and the oracle's Huff0 reader turns them back into the Sprintz streams."""
    import torch
    chunk_len, ndims = 5120, 8
    g = torch.Generator(device="cuda:0").manual_seed(7 + nchunks)
    steps = torch.randint(-8, 9, (nchunks, chunk_len // ndims, ndims), device="cuda:0", generator=g, dtype=torch.int32)
    steps[:, 300:340] = 0
    x = (torch.cumsum(steps, dim=1) + 30000).to(torch.int16).view(torch.uint16).reshape(-1)
    del steps
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
    assert int(bo[-1].item()) < int(oo[-1].item())                    # the entropy stage shrinks this data
    # a sample of whole segments against the specification of the writer and the oracle's reader
    comp, offs, sz_h = batch.data.cpu().numpy(), batch.offsets.cpu().numpy().astype(np.uint64), batch.sizes.cpu().numpy()
    bo_h, blk_h = bo.cpu().numpy(), blocks.cpu().numpy()
    for seg in range(0, nchunks // 64, max(1, nchunks // 64 // 6)):
        c0 = 64 * seg
        want, wo = oracle.huf0_compress(comp, offs[c0:c0 + 65], sz_h[c0:c0 + 64])
        got = blk_h[bo_h[c0]:bo_h[c0 + 64]]
        assert np.array_equal(got, want), seg
        for c in (c0, c0 + 17, c0 + 63):
            plain, ret = oracle.huf0_decompress(blk_h[bo_h[c]:bo_h[c + 1]], int(sz_h[c]))
            assert ret == sz_h[c] and np.array_equal(plain, comp[int(offs[c]):int(offs[c]) + int(sz_h[c])]), c


def test_writer_edge_chunks(sz, oracle):
    """any byte container goes in: empty, 1-byte, 11/12/13-byte chunks, one repeated byte, incompressible
    noise, streams too long for the 16-bit jump table (stored), a chunk count that is not a multiple of 64"""
    import torch
    from sprintz_amd import _lib
    rng = np.random.default_rng(77)
    chunks = []
    for n in (0, 1, 2, 11, 12, 13, 40, 300, 4096, 70000, 300000):
        for k in (1, 2, 5, 60, 256):
            p = 1.0 / np.arange(1, k + 1) ** 1.5
            chunks.append(rng.choice(k, n, p=p / p.sum()).astype(np.uint8))
    chunks.append(rng.integers(0, 256, 5000).astype(np.uint8))
    chunks.append(np.full(9000, 7, np.uint8))
    order = rng.permutation(len(chunks))
    chunks = [chunks[i] for i in order] * 3                                   # 171 chunks: 2 full segments + a partial one
    n = len(chunks)
    sizes = np.array([c.size for c in chunks], np.uint32)
    offs = np.zeros(n + 1, np.uint64)
    offs[1:] = np.cumsum((sizes.astype(np.int64) + 15) & ~15)               # 16-byte aligned starts, like the codec's container
    dense = np.zeros(int(offs[-1]) + 16, np.uint8)
    for c, o in zip(chunks, offs[:-1]):
        dense[int(o):int(o) + c.size] = c
    want, wo = oracle.huf0_compress(dense, offs, sizes)
    d_dense = torch.from_numpy(dense).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    d_sizes = torch.from_numpy(sizes.view(np.int32)).cuda()
    blocks = torch.full((int(_lib.huf0_bound(int(sizes.sum()), n)) + 64,), 0xEE, dtype=torch.uint8, device="cuda")
    bo = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    tmp = torch.empty(int(_lib.huf0_tmp_bytes(n)), dtype=torch.uint8, device="cuda")
    _lib.check(_lib.huf0_compress_batch(d_dense.data_ptr(), d_offs.data_ptr(), d_sizes.data_ptr(), n, blocks.data_ptr(), bo.data_ptr(),
                                        tmp.data_ptr(), None))
    torch.cuda.synchronize()
    assert np.array_equal(bo.cpu().numpy().astype(np.uint64), wo)
    got = blocks.cpu().numpy()
    assert np.array_equal(got[: want.size], want)
    assert (got[want.size:] == 0xEE).all()                                    # nothing written past the last block
    kinds = {"stored": 0, "one": 0, "coded": 0}
    for c in range(n):
        b = int(wo[c + 1] - wo[c])
        if sizes[c]:
            kinds["stored" if b == sizes[c] else "one" if b == 1 else "coded"] += 1
    assert min(kinds.values()) > 5, kinds
    # and back
    oo = np.zeros(n + 1, np.int64)
    oo[1:] = np.cumsum(sizes.astype(np.int64))
    rets = torch.empty(n, dtype=torch.int64, device="cuda")
    out = sz.huf0_decompress(blocks, bo, torch.from_numpy(oo).cuda(), rets=rets).cpu().numpy()
    assert np.array_equal(rets.cpu().numpy(), sizes.astype(np.int64))
    assert np.array_equal(out[: oo[-1]], np.concatenate(chunks))


def test_reader_survives_garbage(sz, oracle, golden_huf0):
    """2 000 blocks damaged in the tree description (where the FSE parser lives), the jump table or anywhere, and
    blocks of pure noise: the launch ends, undamaged neighbours decode, verdicts agree with the oracle"""
    import torch
    manifest, arrays = golden_huf0
    rng = np.random.default_rng(99)
    coded = [m for m in manifest if m["kind"] in ("fse", "nibbles") and m["n"] <= 4200]
    blocks, plains, clean = [], [], []
    for k in range(2000):
        m = coded[int(rng.integers(0, len(coded)))]
        plain, blk = arrays["p%04d" % m["idx"]], arrays["b%04d" % m["idx"]].copy()
        mode = k % 5
        if mode == 0:
            clean.append(k)
        elif mode == 1:
            for _ in range(int(rng.integers(1, 4))):
                blk[rng.integers(0, min(blk.size, blk[0] + 2 if blk[0] < 128 else 70))] = rng.integers(0, 256)
        elif mode == 2:
            hl = (blk[0] + 1) if blk[0] < 128 else (1 + (blk[0] - 126) // 2)
            at = min(int(hl) + int(rng.integers(0, 6)), blk.size - 1)
            blk[at] = rng.integers(0, 256)
        elif mode == 3:
            blk[rng.integers(0, blk.size)] ^= 1 << rng.integers(0, 8)
        else:
            blk = rng.integers(0, 256, int(rng.integers(2, max(3, plain.size - 1)))).astype(np.uint8)
        blocks.append(blk)
        plains.append(plain)
    d, bo, oo, oo_h = pack(blocks, plains)
    rets = torch.full((len(blocks),), -99, dtype=torch.int64, device="cuda")
    out = sz.huf0_decompress(d, bo, oo, rets=rets).cpu().numpy()
    r = rets.cpu().numpy()
    for k in clean:
        assert r[k] == plains[k].size and np.array_equal(out[oo_h[k]:oo_h[k + 1]], plains[k]), k
    agree = rejected = 0
    for k in range(0, len(blocks), 7):                                        # the oracle is slow: a sample
        got, ret = oracle.huf0_decompress(blocks[k], plains[k].size)
        if ret < 0:
            assert r[k] < 0, (k, r[k])
            rejected += 1
        else:
            assert r[k] == plains[k].size and np.array_equal(out[oo_h[k]:oo_h[k + 1]], got), k
        agree += 1
    assert agree > 250 and rejected > 50

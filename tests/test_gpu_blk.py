"""GPU tests (-m gpu) of the BLOCK-PARALLEL delta kernels (csrc/encode_blk.h, decode_blk.h; SPRINTZ_OPT_BLK_CHUNKS): every
shape family they take, batched through the C-ABI, against the oracle -- stream bytes, sizes, return values, samples.  The
general parity modules run on these kernels too (tests/conftest.py: decode_path "blk"); here the shapes are chosen to sit
ON the new kernels (the module asserts that the option is honoured by comparing with the lane-per-column kernels' bytes)."""
import numpy as np
import pytest

from harness import DTYPES, gen_walk

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sz():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import sprintz_amd
    return sprintz_amd


@pytest.fixture(params=["blk", "row", "old"])
def path(request):
    """every test on the block-parallel kernels ("blk": encode_blk + decode_blk + encode_blk_uni; "row": encode_blk + the piece-sequential
    decoder decode_row.h) and, as a control, on the kernels they replace (same bytes)"""
    import os
    from sprintz_amd import _lib
    _lib.check(_lib.set_option(_lib.OPT_LAT_CHUNKS, 0))
    _lib.check(_lib.set_option(_lib.OPT_BLK_CHUNKS, 0 if request.param == "old" else 1))
    _lib.check(_lib.set_option(_lib.OPT_BLK_KERNELS, 25 if request.param == "row" else 7))
    yield request.param
    _lib.set_option(_lib.OPT_LAT_CHUNKS, int(os.environ.get("SPRINTZ_MI355X_LAT_CHUNKS", 2048)))
    _lib.set_option(_lib.OPT_BLK_CHUNKS, int(os.environ.get("SPRINTZ_MI355X_BLK_CHUNKS", 2049)))
    _lib.set_option(_lib.OPT_BLK_KERNELS, int(os.environ.get("SPRINTZ_MI355X_BLK_KERNELS", 9)))


def make_data(kind, rng, n, ndims, esz):
    if kind == "walk":
        return gen_walk(rng, n, ndims, esz, 3)
    if kind == "walk_flat":                                   # flat spans: runs inside and across groups, runs that end a chunk
        return gen_walk(rng, n, ndims, esz, 5, flat_every=2)
    if kind == "zeros":
        return np.zeros(n, DTYPES[esz])
    if kind == "const_cols":                                  # all-zero deltas after the first row: one long run per chunk
        return np.tile(rng.integers(0, 1 << (8 * esz), ndims).astype(DTYPES[esz]), (n + ndims - 1) // ndims)[:n]
    if kind == "noise":                                       # every width at its maximum
        return rng.integers(0, 1 << (8 * esz), n).astype(DTYPES[esz])
    if kind == "mixed":                                       # per column a different step size: every width 0 .. W in one row
        rows = (n + ndims - 1) // ndims
        amp = (1 << (rng.integers(0, 8 * esz + 1, ndims))) >> 1
        steps = rng.integers(-1, 2, (rows, ndims)) * amp[None, :]
        steps[(np.arange(rows) // 24) % 3 == 1] = 0
        return np.mod(np.cumsum(steps, axis=0), 1 << (8 * esz)).astype(DTYPES[esz]).ravel()[:n]
    raise ValueError(kind)


SHAPES = [
    # esz, ndims, chunk_len (elements), nchunks, ragged last chunk (elements short of a full one)
    (1, 80, 10240, 37, 0),            # BASELINE config 3 at 10 KB
    (1, 80, 10240, 5, 3000),
    (1, 16, 2048, 64, 0),
    (1, 16, 16 * 16 * 3 + 32, 19, 48),  # chunk not a whole number of blocks
    (1, 32, 4096, 33, 0),
    (1, 48, 48 * 40, 21, 0),
    (1, 64, 8192, 17, 64 * 5),
    (1, 96, 96 * 24, 13, 0),
    (1, 128, 128 * 16, 11, 0),        # exactly one group a chunk
    (1, 256, 256 * 16 * 2, 6, 0),
    (2, 8, 5120, 70, 0),              # the headline shape on the delta codec
    (2, 8, 5120, 9, 1000),
    (2, 8, 8 * 16 * 2 + 8, 40, 0),
    (2, 16, 4096, 21, 0),
    (2, 24, 24 * 56, 15, 24 * 3),
    (2, 40, 40 * 128, 9, 0),
    (2, 64, 64 * 32, 9, 0),
    (2, 80, 80 * 64, 7, 0),
    (2, 128, 128 * 16, 5, 0),
    # rows of whole dwords that are not whole 16-byte pieces: the column-group-sequential decoder alone (decode_row.h)
    (1, 12, 12 * 40, 50, 0),
    (1, 20, 20 * 64, 31, 20 * 7),
    (1, 8, 1024, 90, 0),
    (2, 6, 6 * 80, 41, 0),
    (2, 10, 10 * 48, 33, 0),
    (2, 4, 2048, 60, 0),
    # univariate streams of the low-dim layout (encode_blk_uni_kernel)
    (1, 1, 1024, 300, 0),             # BASELINE config 1
    (1, 1, 1024, 67, 500),
    (1, 1, 4096, 40, 16),
    (1, 1, 272, 90, 0),
    (1, 1, 128, 90, 0),               # the shortest coded chunk
    (1, 1, 112, 33, 0),               # below 128 elements: verbatim
    (2, 1, 512, 150, 0),
    (2, 1, 2048, 40, 200),
    (2, 1, 200, 77, 0),
]


@pytest.mark.parametrize("kind", ["walk", "walk_flat", "zeros", "const_cols", "noise", "mixed"])
@pytest.mark.parametrize("esz,ndims,chunk_len,nchunks,short", SHAPES)
def test_delta_batches_match_the_oracle(sz, oracle, path, kind, esz, ndims, chunk_len, nchunks, short):
    import torch
    rng = np.random.default_rng(1000 * ndims + chunk_len + esz)
    n = nchunks * chunk_len - short
    data = make_data(kind, rng, n, ndims, esz)
    cd = sz.ChunkedCodec("delta", esz, ndims, chunk_len, device="cuda:0")
    t = torch.from_numpy(data.view(np.int8 if esz == 1 else np.int16)).cuda().view(cd.dtype)
    batch = cd.compress(t)
    sizes, offs, comp = batch.sizes.cpu().numpy(), batch.offsets.cpu().numpy(), batch.data.cpu().numpy()
    for c in range(nchunks):
        want, wret = oracle.compress("delta", data[c * chunk_len:(c + 1) * chunk_len], ndims)
        assert sizes[c] == want.size, (path, kind, c, int(sizes[c]), want.size)
        got = comp[offs[c]:offs[c] + sizes[c]]
        if not np.array_equal(got, want):
            bad = np.flatnonzero(got != want)
            raise AssertionError((path, kind, "chunk", c, "first differing stream bytes", bad[:6].tolist(), "of", want.size,
                                  got[bad[:6]].tolist(), want[bad[:6]].tolist()))
    rets = torch.empty(nchunks, dtype=torch.int64, device="cuda:0")
    out = cd.decompress(batch, rets=rets).cpu().numpy().view(DTYPES[esz])
    assert np.array_equal(out[:n], data), (path, kind)
    r = rets.cpu().numpy()
    assert (r[:-1] == chunk_len).all() and r[-1] == chunk_len - short, (path, kind, r[-3:])


@pytest.mark.parametrize("esz,ndims,chunk_len", [(1, 80, 10240), (2, 8, 5120), (1, 16, 1024), (1, 1, 1024), (2, 1, 1024)])
def test_a_large_batch_takes_the_new_kernels_by_default(sz, oracle, esz, ndims, chunk_len):
    """default options, more chunks than SPRINTZ_OPT_BLK_CHUNKS' default: bytes against the oracle on every chunk"""
    import torch
    nchunks = 5000
    rng = np.random.default_rng(7)
    data = gen_walk(rng, nchunks * chunk_len, ndims, esz, 4, flat_every=5)
    cd = sz.ChunkedCodec("delta", esz, ndims, chunk_len, device="cuda:0")
    t = torch.from_numpy(data.view(np.int8 if esz == 1 else np.int16)).cuda().view(cd.dtype)
    batch = cd.compress(t)
    sizes, offs, comp = batch.sizes.cpu().numpy(), batch.offsets.cpu().numpy(), batch.data.cpu().numpy()
    want, stride, wsizes = oracle.compress_chunks_mt("delta", data, chunk_len, ndims)
    assert np.array_equal(sizes, wsizes)
    for c in range(nchunks):
        assert np.array_equal(comp[offs[c]:offs[c] + sizes[c]], want[c * stride:c * stride + sizes[c]]), c
    assert np.array_equal(cd.decompress(batch).cpu().numpy().view(DTYPES[esz]), data)


@pytest.mark.parametrize("esz,ndims,chunk_len,nchunks", [(1, 80, 10240, 23), (2, 8, 5120, 41), (1, 16, 2048, 50), (2, 40, 40 * 64, 17)])
def test_byte_dense_containers(sz, oracle, path, esz, ndims, chunk_len, nchunks):
    """a container whose streams start at ANY byte (sprintz_mi355x_compact with align = 1): the block-parallel decoder stages a stream from the
    16-byte piece that holds its first byte and carries the phase through every bit address"""
    import torch
    rng = np.random.default_rng(31)
    data = gen_walk(rng, nchunks * chunk_len, ndims, esz, 6, flat_every=3)
    cd = sz.ChunkedCodec("delta", esz, ndims, chunk_len, device="cuda:0", align=1)
    t = torch.from_numpy(data.view(np.int8 if esz == 1 else np.int16)).cuda().view(cd.dtype)
    batch = cd.compress(t)
    offs, sizes, comp = batch.offsets.cpu().numpy(), batch.sizes.cpu().numpy(), batch.data.cpu().numpy()
    assert (np.diff(offs) == sizes).all() and (offs % 16 != 0).any()
    for c in range(nchunks):
        want, _ = oracle.compress("delta", data[c * chunk_len:(c + 1) * chunk_len], ndims)
        assert np.array_equal(comp[offs[c]:offs[c + 1]], want), (path, c)
    rets = torch.empty(nchunks, dtype=torch.int64, device="cuda:0")
    out = cd.decompress(batch, rets=rets).cpu().numpy().view(DTYPES[esz])
    assert np.array_equal(out, data), path
    assert (rets.cpu().numpy() == chunk_len).all()


@pytest.mark.parametrize("esz,ndims,chunk_len,nchunks", [(1, 80, 10240, 40), (2, 8, 5120, 64), (1, 16, 2048, 70), (2, 40, 40 * 64, 24)])
def test_damaged_streams_stay_inside_their_chunk(sz, path, esz, ndims, chunk_len, nchunks):
    """bit-flipped / zeroed / all-ones / header-damaged streams through the new decoders: they terminate, write nothing outside their
    chunk's slot and report SPRINTZ_E_CORRUPT or a count within the chunk"""
    import torch
    rng = np.random.default_rng(77)
    data = gen_walk(rng, nchunks * chunk_len, ndims, esz, 8, flat_every=4)
    cd = sz.ChunkedCodec("delta", esz, ndims, chunk_len, device="cuda:0")
    batch = cd.compress(torch.from_numpy(data.view(np.int8 if esz == 1 else np.int16)).cuda().view(cd.dtype))
    comp0 = batch.data.cpu().numpy().copy()
    offs = batch.offsets.cpu().numpy()
    for trial in range(7):
        comp = comp0.copy()
        if trial == 0:
            for c in range(nchunks):
                comp[offs[c]:offs[c] + 4] = 0xFF
        elif trial == 1:
            comp[:] = 0
        elif trial == 2:
            comp[:] = 0xFF
        elif trial == 3:                                     # truncated: every stream's tail zeroed
            for c in range(nchunks):
                comp[(offs[c] + offs[c + 1]) // 2:offs[c + 1]] = 0
        else:
            idx = rng.integers(0, comp.size, comp.size // 50)
            comp[idx] ^= rng.integers(1, 256, idx.size).astype(np.uint8)
        guard = 4096
        out = torch.full((nchunks * chunk_len + guard,), 0x5A, dtype=torch.int16 if esz == 2 else torch.int8, device="cuda:0")
        rets = torch.zeros(nchunks, dtype=torch.int64, device="cuda:0")
        cd.decompress_into(torch.from_numpy(comp).cuda(), batch.offsets, nchunks, out, rets)
        torch.cuda.synchronize()
        r = rets.cpu().numpy()
        assert ((r == sz._lib.E_CORRUPT) | ((r >= 0) & (r <= chunk_len))).all(), (path, trial, r[:8])
        assert (out[nchunks * chunk_len:].cpu().numpy() == 0x5A).all(), (path, trial)

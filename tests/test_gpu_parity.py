"""GPU parity tests (-m gpu): the HIP path, called through the C-ABI, against
the oracle and the golden vectors.  Bar: bit-exact stream bytes, byte lengths,
return values and decoded samples.  Nothing here reads /root/reference."""
import zlib

import os

import numpy as np
import pytest

from harness import DTYPES, REF_TEST_SIZES, gen_fuzz, gen_patterns, gen_sparse, gen_walk

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("decode_path")]


@pytest.fixture(scope="module")
def sz():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import sprintz_amd
    return sprintz_amd


def gpu_compress(sz, codec, data, ndims, write_size=True):
    esz = data.dtype.itemsize
    fn = getattr(sz, f"sprintz_compress_{codec}_{8 * esz}b")
    dest = np.full((data.size * 3 // 2 + 64) * esz + 8 * ndims + 256, 0xAB, np.uint8)
    ret = fn(data, data.size, dest, ndims, write_size)
    return dest, ret


def gpu_decompress(sz, codec, stream, esz, n):
    fn = getattr(sz, f"sprintz_decompress_{codec}_{8 * esz}b")
    dest = np.full(n + 64, 0xCD if esz == 1 else 0xCDCD, DTYPES[esz])
    ret = fn(np.ascontiguousarray(stream), dest)
    return dest, ret


# ------------------------------------------------------------- single-call API

def test_golden_vectors_single_call(sz, golden):
    """every golden case: GPU encoder output == reference bytes/return value, GPU decoder inverts it"""
    manifest, arrays = golden
    for m in manifest:
        data = arrays[f"in_{m['idx']}"]
        want = arrays[f"out_{m['idx']}"]
        dest, ret = gpu_compress(sz, m["codec"], data, m["ndims"])
        assert ret == m["ret"], (m, sz.last_error())
        assert np.array_equal(dest[:want.size], want), m
        assert (dest[want.size:] == 0xAB).all(), "wrote outside the stream (the caller's buffer holds exactly the stream afterwards)"
        dec, dret = gpu_decompress(sz, m["codec"], want, m["esz"], data.size)
        assert dret == data.size, (m, sz.last_error())
        assert np.array_equal(dec[:data.size], data), m
        assert (dec[data.size:] == dec[-1]).all(), "decoder wrote past the decoded length"


def test_golden_vectors_of_513_to_2047_columns(sz, golden_wide):
    """the reference's own streams at 513 .. 2 047 columns (golden_wide_v1; csrc/any_ndims.hip): the single-call encoder writes them,
    the decoder inverts them, and so does the batched pair on the same samples as one chunk"""
    import torch
    manifest, arrays = golden_wide
    for m in manifest:
        data, want = arrays[f"in_{m['idx']}"], arrays[f"out_{m['idx']}"]
        dest, ret = gpu_compress(sz, m["codec"], data, m["ndims"])
        assert ret == m["ret"] and np.array_equal(dest[:want.size], want), (m, sz.last_error())
        dec, dret = gpu_decompress(sz, m["codec"], want, m["esz"], data.size)
        assert dret == data.size and np.array_equal(dec[:data.size], data.ravel()), m
        cd = sz.ChunkedCodec(m["codec"], m["esz"], m["ndims"], data.size, device="cuda:0")
        t = torch.from_numpy(data.ravel().view(np.int8 if m["esz"] == 1 else np.int16)).cuda().view(cd.dtype)
        b = cd.compress(t)
        assert int(b.sizes[0]) == want.size and np.array_equal(b.data.cpu().numpy()[: want.size], want), m
        assert np.array_equal(cd.decompress(b).cpu().numpy().view(data.dtype)[: data.size], data.ravel()), m


def test_golden_vectors_of_2048_to_65535_columns(sz, golden_wide2):
    """the reference's streams at 2 048 .. 65 535 columns (golden_wide_v2; the column-tiled kernels of csrc/any_ndims.hip): the single-call
    encoder writes them -- the 8 192-column ones with their truncated remaining_len included -- and the decoder returns what the
    reference's decoder returned (the whole input; the decodable prefix at 8 192 columns; at 65 535, where the reference's decoder
    corrupts its heap, the whole input); the batched pair on the same samples as one chunk"""
    import torch
    manifest, arrays = golden_wide2
    for m in manifest:
        data, want = arrays[f"in_{m['in_idx']}"], arrays[f"out_{m['idx']}"]
        dest, ret = gpu_compress(sz, m["codec"], data, m["ndims"])
        assert ret == m["ret"], (m, ret, sz.last_error())
        assert np.array_equal(dest[:want.size], want) and (dest[want.size:] == 0xAB).all(), m
        want_ret = m["ref_dret"] if m["ref_dret"] is not None else data.size
        dec, dret = gpu_decompress(sz, m["codec"], want, m["esz"], data.size)
        assert dret == want_ret and np.array_equal(dec[:dret], data.ravel()[:dret]), (m, dret, sz.last_error())
        cd = sz.ChunkedCodec(m["codec"], m["esz"], m["ndims"], data.size, device="cuda:0")
        t = torch.from_numpy(data.ravel().view(np.int8 if m["esz"] == 1 else np.int16)).cuda().view(cd.dtype)
        # the BATCHED entry points refuse a shape whose verbatim tail can outgrow the header's 16-bit remaining_len (api.hip:
        # check_batch_tail -- whatever the data: two blocks when the chunk is whole blocks, else one block + the ragged rest)
        blk = 8 * m["ndims"]
        tail_max = data.size if data.size < 2 * blk else (2 * blk if data.size % blk == 0 else blk + data.size % blk)
        if m["ndims"] >= 4096 and tail_max > 0xffff:
            with pytest.raises(sz.SprintzError, match="remaining_len"):
                cd.compress(t)
            continue
        assert want_ret == data.size, m                          # (every shape the batch accepts is one the format holds in full)
        b = cd.compress(t)
        assert int(b.sizes[0]) == want.size and np.array_equal(b.data.cpu().numpy()[: want.size], want), m
        assert np.array_equal(cd.decompress(b).cpu().numpy().view(data.dtype)[: data.size], data.ravel()), m


@pytest.mark.parametrize("codec,esz,ndims,n", [("xff", 2, 8, 16384), ("xff", 2, 8, 20003), ("delta", 2, 64, 20480), ("xff", 2, 2, 18001), ("delta", 2, 1, 17000),
                                               ("delta", 1, 8, 24000), ("xff", 1, 16, 22005), ("xff", 1, 1, 20000), ("xff", 1, 3, 23999), ("xff", 2, 33, 19999)])
def test_single_calls_of_17_to_40_KB(sz, oracle, codec, esz, ndims, n):
    """single chunks above 16 KB still take the workgroup-per-chunk kernels while one chunk's working set fits a workgroup's LDS
    (api.hip: lat_chunk_fits): the oracle's bytes, the oracle's samples; flat spans put runs across the larger block counts"""
    rng = np.random.default_rng(n)
    for flat in (0, 3):
        data = gen_walk(rng, n, ndims, esz, 8, flat_every=flat)
        want, wret = oracle.compress(codec, data, ndims)
        dest, ret = gpu_compress(sz, codec, data, ndims)
        assert ret == wret and np.array_equal(dest[:want.size], want), (codec, esz, ndims, n, flat, sz.last_error())
        dec, dret = gpu_decompress(sz, codec, want, esz, data.size)
        assert dret == data.size and np.array_equal(dec[:data.size], data.ravel()), (codec, esz, ndims, n, flat)


def test_random_shapes_against_oracle(sz, oracle):
    """both kernel families (decode_path) over random shapes the fixed matrices do not list: every lanes-per-chunk bucket of both
    layouts (1 .. 64 columns, both widths, both codecs), chunk lengths that are and are not whole groups, a ragged last chunk, data
    with noise, runs (some longer than 127 blocks), constant columns and full-width fields -- stream bytes, sizes and samples the oracle's"""
    import torch
    rng = np.random.default_rng(20240929)
    for trial in range(70):
        esz = int(rng.choice([1, 2]))
        ndims = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 15, 16, 17, 24, 31, 32, 33, 48, 63, 64]))
        codec = str(rng.choice(["delta", "xff"]))
        rows = int(rng.integers(3, 1 + max(4, min(2000, (16384 // esz) // ndims))))
        chunk_len = rows * ndims
        if rng.random() < 0.5:
            chunk_len = max(16, (chunk_len * esz // 16) * 16 // esz)          # 16-byte multiples take the one-workgroup-per-chunk kernels
        nchunks = int(rng.integers(1, 7))
        total = nchunks * chunk_len - int(rng.integers(0, max(1, chunk_len // 2)))
        kind = trial % 5
        nrows = (total + ndims - 1) // ndims
        if kind == 0:
            x = gen_walk(rng, total, ndims, esz, int(rng.integers(1, 40)), flat_every=int(rng.integers(0, 4)))
        elif kind == 1:
            x = rng.integers(0, 1 << (8 * esz), size=total).astype(DTYPES[esz])                       # full-width fields
        elif kind == 2:                                                                                 # long runs: constant, then a step
            m = np.zeros((nrows, ndims), np.int64) + rng.integers(0, 200, size=(1, ndims))
            m[nrows // 2:] += 3
            x = np.mod(m, 1 << (8 * esz)).astype(DTYPES[esz]).ravel()[:total]
        elif kind == 3:                                                                                 # some columns constant, some noisy
            m = np.cumsum(rng.integers(-3, 4, size=(nrows, ndims)), axis=0) * (rng.random(ndims) < 0.5)[None, :] + 77
            x = np.mod(m, 1 << (8 * esz)).astype(DTYPES[esz]).ravel()[:total]
        else:                                                                                           # oscillation: the forecast's counters move
            m = (np.arange(nrows)[:, None] % 2) * int(rng.integers(1, 120)) + rng.integers(0, 3, size=(nrows, ndims))
            x = np.mod(m, 1 << (8 * esz)).astype(DTYPES[esz]).ravel()[:total]
        cd = sz.ChunkedCodec(codec, esz, ndims, chunk_len, device="cuda:0")
        t = torch.from_numpy(x.view(np.int8 if esz == 1 else np.int16)).cuda().view(cd.dtype)
        batch = cd.compress(t)
        comp, offs, sizes = batch.data.cpu().numpy(), batch.offsets.cpu().numpy(), batch.sizes.cpu().numpy()
        for c in range(nchunks):
            want, _ = oracle.compress(codec, x[c * chunk_len:min((c + 1) * chunk_len, total)], ndims)
            assert sizes[c] == want.size, (trial, codec, esz, ndims, chunk_len, c)
            assert np.array_equal(comp[offs[c]:offs[c] + sizes[c]], want), (trial, codec, esz, ndims, chunk_len, c)
        rets = torch.empty(nchunks, dtype=torch.int64, device="cuda:0")
        out = cd.decompress(batch, rets=rets).cpu().numpy()
        assert np.array_equal(out.view(DTYPES[esz]), x), (trial, codec, esz, ndims, chunk_len)
        r = rets.cpu().numpy()
        assert (r[:-1] == chunk_len).all() and r[-1] == total - (nchunks - 1) * chunk_len


@pytest.mark.parametrize("codec,esz,ndims", [("xff", 2, 513), ("delta", 1, 600), ("xff", 1, 1000), ("delta", 2, 1000), ("xff", 2, 2047), ("xff", 1, 2047)])
def test_streams_of_513_to_2047_columns(sz, oracle, codec, esz, ndims):
    """the reference takes any uint16 ndims (format.h:36-45; its own tests stop at 129): 513 .. 2047 columns go through
    csrc/any_ndims.hip -- single calls and batches, whole and ragged chunks, noise / runs (> 127 blocks too) / full-width fields,
    stream bytes and samples the oracle's; a damaged stream is rejected or stays inside its slot"""
    import torch
    rng = np.random.default_rng(ndims * 7 + esz)
    for rows, kind in ((8 * 40 + 5, "walk"), (16, "walk"), (8 * 300, "flat"), (48, "uniform"), (100, "walkflat")):
        n = rows * ndims
        if kind == "uniform":
            x = rng.integers(0, 1 << (8 * esz), size=n).astype(DTYPES[esz])
        elif kind == "flat":
            m = np.zeros((rows, ndims), np.int64) + rng.integers(0, 200, size=(1, ndims))
            m[8 * 280:] += rng.integers(0, 5, size=(rows - 8 * 280, ndims))
            x = np.mod(m, 1 << (8 * esz)).astype(DTYPES[esz]).ravel()
        else:
            x = gen_walk(rng, n, ndims, esz, 6, flat_every=3 if kind == "walkflat" else 0)
        want, wret = oracle.compress(codec, x, ndims)
        dest, ret = gpu_compress(sz, codec, x, ndims)
        assert ret == wret, (codec, esz, ndims, rows, kind, sz.last_error())
        assert np.array_equal(dest[:want.size], want), (codec, esz, ndims, rows, kind)
        dec, dret = gpu_decompress(sz, codec, want, esz, n)
        assert dret == n and np.array_equal(dec[:n], x), (codec, esz, ndims, rows, kind)
    # a batch: 5 chunks of 24 rows, the last one ragged
    chunk_len, nchunks = 24 * ndims, 5
    total = nchunks * chunk_len - 3 * ndims - 1
    x = gen_walk(rng, total, ndims, esz, 9, flat_every=2)
    cd = sz.ChunkedCodec(codec, esz, ndims, chunk_len, device="cuda:0")
    t = torch.from_numpy(x.view(np.int8 if esz == 1 else np.int16)).cuda().view(cd.dtype)
    batch = cd.compress(t)
    comp, offs, sizes = batch.data.cpu().numpy(), batch.offsets.cpu().numpy(), batch.sizes.cpu().numpy()
    for c in range(nchunks):
        w, _ = oracle.compress(codec, x[c * chunk_len:min((c + 1) * chunk_len, total)], ndims)
        assert sizes[c] == w.size and np.array_equal(comp[offs[c]:offs[c] + sizes[c]], w), (codec, esz, ndims, c)
    rets = torch.empty(nchunks, dtype=torch.int64, device="cuda:0")
    out = cd.decompress(batch, rets=rets).cpu().numpy()
    assert np.array_equal(out.view(DTYPES[esz]), x)
    assert rets.cpu().numpy().tolist() == [chunk_len] * (nchunks - 1) + [total - (nchunks - 1) * chunk_len]
    # damage: flipped bits and a cut stream -- E_CORRUPT or a count inside the slot, the guard words untouched
    bad = comp.copy()
    idx = rng.integers(0, bad.size - 16, bad.size // 60)
    bad[idx] ^= rng.integers(1, 256, idx.size).astype(np.uint8)
    guard = 4096
    o2 = torch.full((nchunks * chunk_len + guard,), 0x5A, dtype=torch.int16 if esz == 2 else torch.int8, device="cuda:0")
    r2 = torch.zeros(nchunks, dtype=torch.int64, device="cuda:0")
    cd.decompress_into(torch.from_numpy(bad).cuda(), batch.offsets, nchunks, o2, r2)
    r = r2.cpu().numpy()
    assert ((r == sz._lib.E_CORRUPT) | ((r >= 0) & (r <= chunk_len))).all(), r
    assert (o2[nchunks * chunk_len:].cpu().numpy() == 0x5A).all()


def test_reference_decoder_quirk_on_request(sz, oracle, golden, request):
    """SPRINTZ_OPT_REF_DECODER_QUIRK: the decoders replay the runs of 16-bit general-layout FIRE streams as the REFERENCE DECODER
    does (sprintz_xff_rle.cpp:893-901) -- sample for sample what the compiled reference returned for the golden stream it does
    not invert (tests/golden/golden_refdec_v1), and what the oracle's model of it returns on batches with runs; every other
    codec / width / layout is untouched by the switch"""
    import json
    import torch
    from sprintz_amd import _lib
    _lib.check(_lib.set_option(_lib.OPT_REF_DECODER_QUIRK, 1))
    request.addfinalizer(lambda: _lib.set_option(_lib.OPT_REF_DECODER_QUIRK, 0))
    manifest, arrays = golden
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    cases = json.load(open(os.path.join(gdir, "golden_refdec_v1.json")))["cases"]
    refdec = np.load(os.path.join(gdir, "golden_refdec_v1.npz"))
    for c in cases:
        m = [mm for mm in manifest if mm["idx"] == c["idx"]][0]
        data, stream = arrays[f"in_{m['idx']}"], arrays[f"out_{m['idx']}"]
        dec, dret = gpu_decompress(sz, m["codec"], stream, m["esz"], data.size)
        assert dret == c["dec_ret"]
        assert np.array_equal(dec[:data.size], refdec[f"refdec_{m['idx']}"]), m["name"]
        assert not np.array_equal(dec[:data.size], data.ravel())
    rng = np.random.default_rng(41)

    def oscillate_then_run(nchunks, rows, ndims):
        """golden_v1's fire16_run_nonzero_pred over and over: an oscillation (the counters go negative), the decay FIRE predicts,
        48 constant rows (RUN blocks that start with a non-zero prediction), noise"""
        out = np.zeros((nchunks, rows, ndims), np.int64)
        for c in range(nchunks):
            v, seq = 1000 + int(rng.integers(0, 500)), []
            while len(seq) < rows:
                for i in range(8):
                    v += 100 if i % 2 == 0 else -100
                    seq.append([v] * ndims)
                for dl in (6, -1, 0, 0, 0, 0, 0, 0):
                    v += dl
                    seq.append([v] * ndims)
                seq += [[v] * ndims] * 48
                seq += [list(rng.integers(0, 50, ndims) + v) for _ in range(32)]
            out[c] = np.array(seq[:rows])
        return np.mod(out, 65536).astype(np.uint16).reshape(-1)
    for ndims, chunk_len, nchunks in ((3, 3000, 40), (8, 5120, 40), (8, 5120, 2000), (17, 17 * 320, 30), (80, 10240, 24)):
        data = oscillate_then_run(nchunks, chunk_len // ndims, ndims)
        cd = sz.ChunkedCodec("xff", 2, ndims, chunk_len, device="cuda:0")
        batch = cd.compress(torch.from_numpy(data.view(np.int16)).cuda().view(torch.uint16))
        out = cd.decompress(batch).cpu().numpy().view(np.uint16)
        comp, offs, sizes = batch.data.cpu().numpy(), batch.offsets.cpu().numpy(), batch.sizes.cpu().numpy()
        differs = 0
        for c in list(range(0, nchunks, max(1, nchunks // 40))):
            want, _ = oracle.decompress("xff", comp[offs[c]:offs[c] + sizes[c]], 2, chunk_len, quirk=1)
            got = out[c * chunk_len:(c + 1) * chunk_len]
            assert np.array_equal(got, want), (ndims, nchunks, c)
            differs += int(not np.array_equal(got, data[c * chunk_len:(c + 1) * chunk_len]))
        assert differs > 0, (ndims, "the inputs were meant to hit the divergence")
    for codec, esz, ndims in (("delta", 2, 8), ("xff", 1, 8), ("xff", 2, 2)):     # not 16-bit general-layout FIRE: no divergence exists
        d = gen_walk(rng, 20 * 4096, ndims, esz, 8, flat_every=3)
        cd = sz.ChunkedCodec(codec, esz, ndims, 4096, device="cuda:0")
        t = torch.from_numpy(d.view(np.int8 if esz == 1 else np.int16)).cuda().view(cd.dtype)
        assert torch.equal(cd.decompress(cd.compress(t)).view(torch.uint8), t.view(torch.uint8)), (codec, esz, ndims)


@pytest.mark.parametrize("wait_mode", [0, 1, 2])
def test_single_calls_from_many_threads(sz, oracle, wait_mode, request):
    """the drop-in symbols are re-entrant (sprintz.h: one call = one thread, any number at once): 24 host threads, each with its own
    shape and data, compress and decompress through the single-call entry points at the same time -- the threads share the
    library's stream pool and wait on their own events (spin / sleep / by the number of callers) -- and every stream and every
    sample is the oracle's"""
    import threading
    from sprintz_amd import _lib
    _lib.check(_lib.set_option(_lib.OPT_HOST_WAIT, wait_mode))
    request.addfinalizer(lambda: _lib.set_option(_lib.OPT_HOST_WAIT, 0))
    shapes = [("xff", 2, 8, 5120), ("delta", 1, 80, 10240), ("xff", 1, 1, 1024), ("delta", 2, 3, 3000), ("xff", 2, 32, 5120), ("xff", 1, 16, 4096)]
    nthreads, rounds = 24, 12
    errors = []

    def work(k):
        try:
            codec, esz, ndims, n = shapes[k % len(shapes)]
            rng = np.random.default_rng(1000 + k)
            for r in range(rounds):
                data = gen_walk(rng, n - (r % 3) * ndims, ndims, esz, 8, flat_every=4 if r % 2 else 0)
                want, wret = oracle.compress(codec, data, ndims)
                dest, ret = gpu_compress(sz, codec, data, ndims)
                assert ret == wret and np.array_equal(dest[:want.size], want), (k, r, "stream")
                dec, dret = gpu_decompress(sz, codec, want, esz, data.size)
                assert dret == data.size and np.array_equal(dec[:data.size], data.ravel()), (k, r, "samples")
        except BaseException as e:                      # noqa: BLE001 -- reported by the main thread
            errors.append((k, repr(e)))
    ths = [threading.Thread(target=work, args=(k,)) for k in range(nthreads)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not errors, errors[:3]


@pytest.mark.parametrize("esz", [1, 2])
@pytest.mark.parametrize("codec", ["delta", "xff"])
def test_reference_test_matrix_single_call(sz, oracle, codec, esz):
    """sizes and input families of the reference's test_codec (compress_testing.hpp:452-486)"""
    rng = np.random.default_rng(123)
    for ndims in [1, 2, 3, 4, 5, 7, 8, 9, 16, 17, 33, 64, 65, 80, 129]:
        for n in REF_TEST_SIZES:
            cases = [gen_fuzz(rng, n, esz, sh) for sh in (0, 3, 6) if sh < 8 * esz]
            cases.append(gen_sparse(rng, n, esz, 0.05))
            cases.append(gen_walk(rng, n, ndims, esz, 8, flat_every=3))
            if ndims in (1, 8, 129):
                cases += [d for _, d in gen_patterns(n, esz)]
            for d in cases:
                want, wret = oracle.compress(codec, d, ndims)
                dest, ret = gpu_compress(sz, codec, d, ndims)
                assert ret == wret, (codec, esz, ndims, n, sz.last_error())
                assert np.array_equal(dest[:want.size], want), (codec, esz, ndims, n)
                dec, dret = gpu_decompress(sz, codec, want, esz, n)
                assert dret == n and np.array_equal(dec[:n], d), (codec, esz, ndims, n)


def test_write_size_false_and_headerless_decode(sz, oracle):
    rng = np.random.default_rng(7)
    for codec in ("delta", "xff"):
        for esz in (1, 2):
            for ndims in (1, 3, 8, 17):
                n = 16 * ndims * 6 + 5
                d = gen_walk(rng, n, ndims, esz, 8)
                want, wret = oracle.compress(codec, d, ndims, write_size=False)
                dest, ret = gpu_compress(sz, codec, d, ndims, write_size=False)
                assert ret == wret and np.array_equal(dest[:want.size], want)
                full, _ = oracle.compress(codec, d, ndims)
                ngroups = int(np.frombuffer(full[:4].tobytes(), np.uint32)[0])
                remaining = int(np.frombuffer(full[4:6].tobytes(), np.uint16)[0])
                out = np.zeros(n + 64, DTYPES[esz])
                r = sz.decompress_noheader(codec, esz, want, out, ndims, ngroups, remaining)
                assert r == n and np.array_equal(out[:n], d)


def test_long_runs(sz, oracle):
    """> 127 blocks (2-byte varint) and > 32767 blocks (run cap, sprintz_xff_rle.cpp:71,455)"""
    for esz, codec, nd in [(1, "delta", 5), (2, "xff", 8), (1, "xff", 2), (2, "delta", 1), (1, "xff", 1)]:
        for nblocks in (130, 32767 + 5):
            n = nblocks * 8 * nd + 3
            d = np.zeros(n, DTYPES[esz])
            d[-2:] = 7
            want, wret = oracle.compress(codec, d, nd)
            dest, ret = gpu_compress(sz, codec, d, nd)
            assert ret == wret and np.array_equal(dest[:want.size], want)
            dec, dret = gpu_decompress(sz, codec, want, esz, n)
            assert dret == n and np.array_equal(dec[:n], d)


def test_decoder_is_lossless_where_reference_decoder_is_not(sz, golden):
    manifest, arrays = golden
    m = [m for m in manifest if not m["ref_roundtrips"]][0]
    data, stream = arrays[f"in_{m['idx']}"], arrays[f"out_{m['idx']}"]
    dec, dret = gpu_decompress(sz, "xff", stream, 2, data.size)
    assert dret == data.size and np.array_equal(dec[:data.size], data)


# ------------------------------------------------------------------ batched API

CONFIGS = [
    # (name, codec, esz, ndims, chunk_len)      BASELINE.json configs, reduced batch
    ("cfg1", "delta", 1, 1, 1024),
    ("cfg2", "xff", 2, 8, 5120),
    ("cfg3_1k", "delta", 1, 80, 1024),          # raw passthrough in the reference (BASELINE.md finding)
    ("cfg3_10k", "delta", 1, 80, 10240),
    ("cfg5", "xff", 2, 32, 5120),
    ("delta16", "delta", 2, 8, 5120),
    ("xff8", "xff", 1, 8, 4096),
    ("odd_d", "xff", 2, 5, 5000),               # chunk not a multiple of the block: ragged tails, scalar store path
    ("lowdim16", "xff", 2, 2, 4096),
    ("lowdim8", "xff", 1, 3, 3000),
    ("uni16_xff", "xff", 2, 1, 2048),           # univariate: decode_uni.h
    ("uni8_xff_ragged", "xff", 1, 1, 1003),
    ("uni16_delta_ragged", "delta", 2, 1, 777),
    ("uni8_tiny", "delta", 1, 1, 100),          # chunks below 128 elements: verbatim
    ("low8_d2", "delta", 1, 2, 2048),           # 2 and 4 columns per lane in decode_uni.h / encode_uni.h
    ("low8_d2_xff_ragged", "xff", 1, 2, 1006),
    ("low8_d4", "xff", 1, 4, 4000),
    ("low8_d4_delta_ragged", "delta", 1, 4, 1500),
    ("low16_d2_delta_ragged", "delta", 2, 2, 1002),
    ("low8_d3_delta", "delta", 1, 3, 3072),      # 24-byte blocks: spans of three 64-byte windows
    ("low8_d3_xff_ragged", "xff", 1, 3, 1000),
    ("wide8_d128_xff", "xff", 1, 128, 16384),    # two columns per lane: encode_wide.h
    ("wide8_d100_delta", "delta", 1, 100, 8000),
    ("wide8_d66_xff", "xff", 1, 66, 6336),
    ("wide16_d80_xff", "xff", 2, 80, 10240),     # the same at 16 bits: two 16-byte pieces per lane and block
    ("wide16_d128_delta", "delta", 2, 128, 8192),
    ("wide16_d72_xff", "xff", 2, 72, 72 * 40),
]


@pytest.mark.parametrize("name,codec,esz,ndims,chunk_len", CONFIGS)
def test_batched_matches_oracle(sz, oracle, name, codec, esz, ndims, chunk_len):
    import torch
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    nchunks = 300
    total = nchunks * chunk_len - chunk_len // 3          # ragged last chunk
    parts = [gen_walk(rng, total // 2, ndims, esz, 8, flat_every=4),
             gen_fuzz(rng, total - total // 2, esz, 3)]
    data = np.concatenate(parts)
    cd = sz.ChunkedCodec(codec, esz, ndims, chunk_len, device="cuda:0")
    batch = cd.compress(torch.from_numpy(data).cuda())
    want = oracle.compress_chunks(codec, data, chunk_len, ndims)
    comp, offs, sizes = batch.data.cpu().numpy(), batch.offsets.cpu().numpy(), batch.sizes.cpu().numpy()
    assert batch.nchunks == len(want) == nchunks
    for c in range(nchunks):
        assert sizes[c] == want[c].size, (name, c)
        assert np.array_equal(comp[offs[c]:offs[c] + sizes[c]], want[c]), (name, c)
        assert offs[c] % 16 == 0
    rets = torch.empty(nchunks, dtype=torch.int64, device="cuda:0")
    out = cd.decompress(batch, rets=rets)
    assert np.array_equal(out.cpu().numpy(), data), name
    r = rets.cpu().numpy()
    assert (r[:-1] == chunk_len).all() and r[-1] == total - (nchunks - 1) * chunk_len


@pytest.mark.parametrize("codec,esz,ndims", [("delta", 1, 1), ("xff", 1, 1), ("delta", 2, 1), ("xff", 2, 1), ("delta", 1, 2), ("xff", 1, 3)])
def test_lowdim_worst_case_streams_keep_the_ring_fed(sz, oracle, codec, esz, ndims):
    """decode_uni.h refills a lane's 128-byte ring on a fixed cadence (every 2nd block for 8-bit univariate streams): the
    bound it rests on is the LARGEST block (every field full width).  Chunks that switch between incompressible noise,
    runs and small steps at random points put the lanes of a wave at every phase of that cadence."""
    import torch
    rng = np.random.default_rng(2024 + esz * 10 + ndims)
    chunk_len, nchunks = 1024 * ndims, 700
    hi = 256 if esz == 1 else 65536
    data = np.zeros(nchunks * chunk_len, dtype=np.uint8 if esz == 1 else np.uint16)
    pos = 0
    while pos < data.size:
        n = int(rng.integers(8, 400)) * ndims
        kind = int(rng.integers(0, 4))
        seg = data[pos:pos + n]
        if kind <= 1:
            seg[:] = rng.integers(0, hi, seg.size)                       # full-width fields, block after block
        elif kind == 2:
            seg[:] = data[pos - 1] if pos else 0                         # runs
        else:
            seg[:] = (np.cumsum(rng.integers(-2, 3, seg.size)) + 100) % hi
        pos += n
    cd = sz.ChunkedCodec(codec, esz, ndims, chunk_len, device="cuda:0")
    batch = cd.compress(torch.from_numpy(data).cuda())
    want = oracle.compress_chunks(codec, data, chunk_len, ndims)
    comp, offs, sizes = batch.data.cpu().numpy(), batch.offsets.cpu().numpy(), batch.sizes.cpu().numpy()
    for c in range(nchunks):
        assert sizes[c] == want[c].size and np.array_equal(comp[offs[c]:offs[c] + sizes[c]], want[c]), c
    assert np.array_equal(cd.decompress(batch).cpu().numpy(), data)
    # the same streams byte-dense (every alignment of a stream's start inside its 64-byte piece)
    dense = np.concatenate(want + [np.zeros(64, np.uint8)])
    doffs = np.zeros(nchunks + 1, np.int64)
    doffs[1:] = np.cumsum([w.size for w in want])
    out = torch.empty(nchunks * chunk_len, dtype=torch.uint8 if esz == 1 else torch.int16, device="cuda:0")
    cd.decompress_into(torch.from_numpy(dense).cuda(), torch.from_numpy(doffs).cuda(), nchunks, out)
    assert np.array_equal(out.cpu().numpy().view(data.dtype), data)


def test_batched_decodes_byte_dense_reference_streams(sz, oracle):
    """streams concatenated with no alignment (odd offsets) decode identically"""
    import torch
    rng = np.random.default_rng(11)
    codec, esz, ndims, chunk_len, nchunks = "xff", 2, 8, 5120, 64
    data = gen_walk(rng, nchunks * chunk_len, ndims, esz, 30, flat_every=3)   # runs -> 1-byte varints -> odd sizes
    streams = oracle.compress_chunks(codec, data, chunk_len, ndims)
    offs = np.zeros(nchunks + 1, np.int64)
    offs[1:] = np.cumsum([s.size for s in streams])
    assert any(o % 2 for o in offs)
    comp = np.concatenate(streams + [np.zeros(16, np.uint8)])
    cd = sz.ChunkedCodec(codec, esz, ndims, chunk_len, device="cuda:0")
    out = torch.empty(nchunks * chunk_len, dtype=torch.uint16, device="cuda:0")
    cd.decompress_into(torch.from_numpy(comp).cuda(), torch.from_numpy(offs).cuda(), nchunks, out)
    assert np.array_equal(out.cpu().numpy(), data)


@pytest.mark.parametrize("k", [2, 3, 7])
def test_consecutive_chunks_per_group(sz, oracle, request, k):
    """decode_fast with several consecutive chunks per lane group (ring read-ahead running
    across chunk boundaries): 16-byte aligned container, byte-dense container, ragged tails"""
    import torch
    from sprintz_amd import _lib
    _lib.check(_lib.set_option(_lib.OPT_CHUNKS_PER_GROUP, k))
    request.addfinalizer(lambda: _lib.set_option(_lib.OPT_CHUNKS_PER_GROUP, 1))
    rng = np.random.default_rng(100 + k)
    for codec, esz, ndims, chunk_len, nchunks in [("xff", 2, 8, 5120, 101), ("delta", 1, 16, 4096, 37),
                                                  ("xff", 2, 8, 5000, 50)]:
        data = np.concatenate([gen_walk(rng, nchunks * chunk_len // 2, ndims, esz, 8, flat_every=3),
                               gen_fuzz(rng, nchunks * chunk_len - nchunks * chunk_len // 2, esz, 2)])
        cd = sz.ChunkedCodec(codec, esz, ndims, chunk_len, device="cuda:0")
        batch = cd.compress(torch.from_numpy(data).cuda())
        out = cd.decompress(batch)
        assert np.array_equal(out.cpu().numpy(), data), (codec, k, "aligned")
        streams = oracle.compress_chunks(codec, data, chunk_len, ndims)
        offs = np.zeros(nchunks + 1, np.int64)
        offs[1:] = np.cumsum([s.size for s in streams])
        comp = np.concatenate(streams + [np.zeros(16, np.uint8)])
        out2 = torch.empty(nchunks * chunk_len, dtype=cd.dtype, device="cuda:0")
        cd.decompress_into(torch.from_numpy(comp).cuda(), torch.from_numpy(offs).cuda(), nchunks, out2)
        assert np.array_equal(out2.cpu().numpy(), data), (codec, k, "dense")


@pytest.mark.parametrize("esz", [1, 2])
@pytest.mark.parametrize("codec", ["delta", "xff"])
@pytest.mark.parametrize("ndims", [65, 66, 68, 70, 72, 74, 76, 78, 79, 80])
def test_split_lane_mapping_65_to_80_columns(sz, oracle, request, codec, ndims, esz):
    """8 bits: the SPLIT mapping of decode_fast.h and encode_wide.h (32 lanes a chunk, a pair + a single column per lane; the
    decoder's LDS carve sized for 80 columns); 16 bits: 64 x 2 with the 80-column carve.  Both against what they replace
    (SPRINTZ_OPT_SPLIT_LANES 0): stream bytes against the oracle's, decodes against the data and on the oracle's own byte-dense
    streams -- every even width of the range (odd 8-bit widths take the generic decoder: their blocks are not 16-byte
    multiples), runs, full-width fields, ragged last chunks, chunk lengths just above the fast path's floor."""
    import torch
    from sprintz_amd import _lib
    request.addfinalizer(lambda: _lib.set_option(_lib.OPT_SPLIT_LANES, 1))
    rng = np.random.default_rng(1000 * esz + ndims)
    for rows, nchunks, step in [(128, 41, 3), (56, 70, 60 if esz == 1 else 9000), (136, 19, 1)]:
        chunk_len = rows * ndims
        total = nchunks * chunk_len - 5 * ndims - 3
        data = np.concatenate([gen_walk(rng, total // 2, ndims, esz, step, flat_every=3), gen_fuzz(rng, total - total // 2, esz, 2)])
        cd = sz.ChunkedCodec(codec, esz, ndims, chunk_len, device="cuda:0")
        streams = oracle.compress_chunks(codec, data, chunk_len, ndims)
        offs = np.zeros(nchunks + 1, np.int64)
        offs[1:] = np.cumsum([s.size for s in streams])
        comp = torch.from_numpy(np.concatenate(streams + [np.zeros(16, np.uint8)])).cuda()
        offs_t = torch.from_numpy(offs).cuda()
        for split in (1, 0):
            _lib.check(_lib.set_option(_lib.OPT_SPLIT_LANES, split))
            batch = cd.compress(torch.from_numpy(data).cuda())              # the encoder's split mapping (encode_wide.h) / 64 x 2
            got, goffs, gsizes = batch.data.cpu().numpy(), batch.offsets.cpu().numpy(), batch.sizes.cpu().numpy()
            for c in range(nchunks):
                assert gsizes[c] == streams[c].size, (codec, esz, ndims, rows, split, c)
                assert np.array_equal(got[goffs[c]:goffs[c] + gsizes[c]], streams[c]), (codec, esz, ndims, rows, split, c)
            rets = torch.empty(nchunks, dtype=torch.int64, device="cuda:0")
            out = cd.decompress(batch, rets=rets)
            assert np.array_equal(out.cpu().numpy(), data), (codec, esz, ndims, rows, split, "aligned container")
            r = rets.cpu().numpy()
            assert (r[:-1] == chunk_len).all() and r[-1] == total - (nchunks - 1) * chunk_len
            out2 = torch.full((nchunks * chunk_len,), 0x5a, dtype=cd.dtype, device="cuda:0")
            cd.decompress_into(comp, offs_t, nchunks, out2)
            assert np.array_equal(out2.cpu().numpy()[:total], data), (codec, esz, ndims, rows, split, "byte-dense container")


def test_decoder_rejects_wrong_ndims(sz, oracle):
    import torch
    rng = np.random.default_rng(3)
    data = gen_walk(rng, 4 * 5120, 8, 2, 8)
    cd8 = sz.ChunkedCodec("xff", 2, 8, 5120, device="cuda:0")
    batch = cd8.compress(torch.from_numpy(data).cuda())
    cd16 = sz.ChunkedCodec("xff", 2, 16, 5120, device="cuda:0")
    rets = torch.zeros(4, dtype=torch.int64, device="cuda:0")
    out = torch.zeros(4 * 5120, dtype=torch.uint16, device="cuda:0")
    cd16.decompress_into(batch.data, batch.offsets, 4, out, rets)
    assert (rets.cpu().numpy() == sz._lib.E_CORRUPT).all()
    assert not out.view(torch.int16).any().item()


def test_host_chunked_convenience(sz, oracle):
    rng = np.random.default_rng(5)
    data = gen_walk(rng, 50 * 5120 + 77, 8, 2, 8, flat_every=5)
    comp, offs = sz.compress_chunked("xff", data, 8, 5120)
    want = np.concatenate(oracle.compress_chunks("xff", data, 5120, 8))
    assert np.array_equal(comp, want)
    dec = sz.decompress_chunked("xff", comp, offs, 2, 8, 5120)
    assert np.array_equal(dec, data)


# ------------------------------------------------ BASELINE.json full sizes (properties)

def test_full_size_cfg2_roundtrip_and_size_checksum(sz, oracle):
    """131072 chunks x 10 KB (1.34 GB): encode->decode round trip on the
    GPU, plus the sizes and the stream bytes of EVERY chunk against the oracle (run over the host's cores)."""
    import torch
    codec, esz, ndims, chunk_len, nchunks = "xff", 2, 8, 5120, 131072
    g = torch.Generator(device="cuda:0").manual_seed(123)
    steps = torch.randint(-8, 9, (nchunks, chunk_len // ndims, ndims), device="cuda:0", generator=g, dtype=torch.int32)
    steps[:, 192:256] = 0                                    # a flat span per chunk: RLE runs
    x = (torch.cumsum(steps, dim=1) + 20000).to(torch.int16).view(torch.uint16).reshape(-1)
    del steps
    cd = sz.ChunkedCodec(codec, esz, ndims, chunk_len, device="cuda:0")
    batch = cd.compress(x)
    out = cd.decompress(batch)
    assert torch.equal(out.view(torch.int16), x.view(torch.int16))
    sizes = batch.sizes.cpu().numpy()
    offs = batch.offsets.cpu().numpy()
    comp = batch.data.cpu().numpy()
    # EVERY chunk against the oracle, run over the host's cores (1.34 GB at ~0.3 GB/s a core): all 131 072 sizes, all stream bytes
    want, stride, wsizes = oracle.compress_chunks_mt(codec, x.cpu().numpy().view(np.uint16), chunk_len, ndims)
    assert np.array_equal(sizes, wsizes), np.flatnonzero(sizes != wsizes)[:8]
    assert zlib.crc32(sizes.tobytes()) == zlib.crc32(wsizes.tobytes())
    for c in range(nchunks):
        n = int(sizes[c])
        if not np.array_equal(comp[offs[c]:offs[c] + n], want[c * stride:c * stride + n]):
            raise AssertionError(f"chunk {c}: stream bytes differ from the oracle's")
    assert 2.0 < x.numel() * 2 / sizes.sum() < 6.0


@pytest.mark.parametrize("name,codec,esz,ndims,chunk_len,nchunks,step", [
    ("cfg1", "delta", 1, 1, 1024, 524288, 2),            # 512 MiB of univariate 8-bit streams in 1 KB chunks
    ("cfg3_1k", "delta", 1, 80, 1024, 524288, 2),        # 80 columns, 1 KB chunks: shorter than one group, stored verbatim
    ("cfg3_10k", "delta", 1, 80, 10240, 52429, 2),       # 80 columns, 10 KB chunks
])
def test_full_size_8bit_configs_roundtrip_and_parity_on_every_chunk(sz, oracle, name, codec, esz, ndims, chunk_len, nchunks, step):
    """BASELINE configs 1 and 3 at the sizes bench.py runs them: encode -> decode round trip on the GPU, and the sizes and
    compressed bytes of EVERY chunk against the oracle (run over the host's cores; round 5 compared a 150-chunk sample)."""
    import torch
    g = torch.Generator(device="cuda:0").manual_seed(321)
    rows = chunk_len // ndims
    tail = chunk_len - rows * ndims
    steps = torch.randint(-step, step + 1, (nchunks, rows, ndims), device="cuda:0", generator=g, dtype=torch.int32)
    if rows > 64:
        steps[:, rows // 2:rows // 2 + 24] = 0                # a flat span per chunk: runs
    body = (torch.cumsum(steps, dim=1) + 100).to(torch.uint8).reshape(nchunks, rows * ndims)
    del steps
    if tail:
        body = torch.cat([body, torch.randint(0, 256, (nchunks, tail), device="cuda:0", generator=g, dtype=torch.int32).to(torch.uint8)], dim=1)
    x = body.reshape(-1).contiguous()
    del body
    cd = sz.ChunkedCodec(codec, esz, ndims, chunk_len, device="cuda:0")
    batch = cd.compress(x)
    rets = torch.empty(nchunks, dtype=torch.int64, device="cuda:0")
    out = cd.decompress(batch, rets=rets)
    assert torch.equal(out, x), name
    assert bool((rets == chunk_len).all().item())
    sizes, offs = batch.sizes.cpu().numpy(), batch.offsets.cpu().numpy()
    comp = batch.data.cpu().numpy()
    want, stride, wsizes = oracle.compress_chunks_mt(codec, x.cpu().numpy(), chunk_len, ndims)
    assert np.array_equal(sizes, wsizes), (name, np.flatnonzero(sizes != wsizes)[:8])
    # the container is the streams at 16-byte aligned starts: compare it as one gather instead of 524 288 slices
    w2 = want[: nchunks * stride].reshape(nchunks, stride)
    col = np.arange(stride, dtype=np.int64)
    per = max(1, (1 << 24) // stride)                           # ~16 M stream bytes (128 MB of gather indices) a pass
    for c0 in range(0, nchunks, per):
        c1 = min(nchunks, c0 + per)
        n = sizes[c0:c1].astype(np.int64)
        keep = col[None, :] < n[:, None]
        got = np.zeros((c1 - c0, stride), np.uint8)
        idx = offs[c0:c1].astype(np.int64)[:, None] + col[None, :]
        got[keep] = comp[idx[keep]]
        bad = np.flatnonzero((got != np.where(keep, w2[c0:c1], 0)).any(axis=1))
        assert bad.size == 0, (name, "chunks whose stream bytes differ from the oracle's", (bad[:8] + c0).tolist())


@pytest.mark.parametrize("codec,esz,ndims,chunk_len,nchunks", [("xff", 2, 8, 5120, 64), ("xff", 1, 1, 1024, 300), ("delta", 2, 1, 2000, 300),
                                                               ("xff", 1, 3, 3000, 64), ("xff", 1, 4, 2000, 300), ("delta", 2, 2, 2000, 300),
                                                               ("delta", 1, 80, 10240, 40), ("xff", 1, 72, 72 * 64, 40),      # the split lane mapping (decode_fast.h, SPLIT)
                                                               ("xff", 2, 80, 10240, 24),                                      # 64 x 2 with the 80-column LDS carve
                                                               ("delta", 1, 80, 1024, 100), ("xff", 2, 8, 100, 100)])          # verbatim batches (verbatim_decode_kernel)
def test_corrupt_streams_do_not_hang_or_overrun(sz, oracle, codec, esz, ndims, chunk_len, nchunks):
    """bit-flipped / truncated / header-damaged streams: the decoder must terminate, stay inside
    each chunk's output slot and either decode something or report SPRINTZ_E_CORRUPT"""
    import torch
    rng = np.random.default_rng(77)
    data = gen_walk(rng, nchunks * chunk_len, ndims, esz, 8, flat_every=4)
    cd = sz.ChunkedCodec(codec, esz, ndims, chunk_len, device="cuda:0")
    batch = cd.compress(torch.from_numpy(data).cuda())
    comp0 = batch.data.cpu().numpy().copy()
    offs = batch.offsets.cpu().numpy()
    for trial in range(6):
        comp = comp0.copy()
        if trial == 0:                                   # huge group counts
            for c in range(nchunks):
                comp[offs[c]:offs[c] + 4] = 0xFF
        elif trial == 1:                                 # zeroed streams
            comp[:] = 0
        elif trial == 2:                                 # all ones
            comp[:] = 0xFF
        else:                                            # random bit flips
            idx = rng.integers(0, comp.size, comp.size // 50)
            comp[idx] ^= rng.integers(1, 256, idx.size).astype(np.uint8)
        guard = 4096
        out = torch.full((nchunks * chunk_len + guard,), 0x5A, dtype=torch.int16 if esz == 2 else torch.int8, device="cuda:0")
        rets = torch.zeros(nchunks, dtype=torch.int64, device="cuda:0")
        cd.decompress_into(torch.from_numpy(comp).cuda(), batch.offsets, nchunks, out, rets)
        torch.cuda.synchronize()
        r = rets.cpu().numpy()
        assert ((r == sz._lib.E_CORRUPT) | ((r >= 0) & (r <= chunk_len))).all(), (trial, r[:8])
        assert (out[nchunks * chunk_len:].cpu().numpy() == 0x5A).all(), trial


@pytest.mark.parametrize("codec,esz,ndims,chunk_len,nchunks", [("xff", 2, 8, 5120, 64), ("delta", 1, 80, 10240, 32), ("xff", 1, 1, 1024, 128),
                                                               ("delta", 2, 300, 9600, 16), ("xff", 2, 8, 5001, 32), ("bitpack", 1, 5, 4000, 32)])
def test_generic_decoder_never_reads_past_a_stream(sz, request, codec, esz, ndims, chunk_len, nchunks):
    """decode_kernel.h (D > 256, odd chunk sizes, the bit-packing codec, and every shape under SPRINTZ_OPT_NO_FAST) checks its
    input cursor against the stream's end: truncated streams are SPRINTZ_E_CORRUPT, bit-flipped ones terminate inside
    their own ranges, and the slot padding of short verbatim chunks is zero"""
    import torch
    from sprintz_amd import _lib
    _lib.check(_lib.set_option(_lib.OPT_NO_FAST, 1))
    request.addfinalizer(lambda: _lib.set_option(_lib.OPT_NO_FAST, 0))
    rng = np.random.default_rng(91)
    data = gen_walk(rng, nchunks * chunk_len, ndims, esz, 6, flat_every=5)
    cd = sz.ChunkedCodec(codec, esz, ndims, chunk_len, device="cuda:0")
    batch = cd.compress(torch.from_numpy(data).cuda())
    comp0, offs, sizes = batch.data.cpu().numpy().copy(), batch.offsets.cpu().numpy(), batch.sizes.cpu().numpy()
    assert np.array_equal(cd.decompress(batch).cpu().numpy(), data)
    # every stream cut short by 1 .. 9 bytes and packed densely again: each must be reported, none decoded past its end
    cut = [comp0[offs[c]:offs[c] + sizes[c] - 1 - (c % 9)] for c in range(nchunks)]
    toffs = np.zeros(nchunks + 1, np.int64)
    toffs[1:] = np.cumsum([s.size for s in cut])
    tcomp = np.concatenate(cut + [np.zeros(16, np.uint8)])
    guard = 4096
    out = torch.full((nchunks * chunk_len + guard,), 0x5A, dtype=torch.int16 if esz == 2 else torch.int8, device="cuda:0")
    rets = torch.zeros(nchunks, dtype=torch.int64, device="cuda:0")
    cd.decompress_into(torch.from_numpy(tcomp).cuda(), torch.from_numpy(toffs).cuda(), nchunks, out, rets)
    torch.cuda.synchronize()
    assert (rets.cpu().numpy() == _lib.E_CORRUPT).all(), rets.cpu().numpy()[:8]
    assert (out[nchunks * chunk_len:].cpu().numpy() == 0x5A).all()
    for trial in range(3):                               # random bit flips: terminate, stay inside the slots
        comp = comp0.copy()
        idx = rng.integers(0, comp.size, comp.size // 40)
        comp[idx] ^= rng.integers(1, 256, idx.size).astype(np.uint8)
        cd.decompress_into(torch.from_numpy(comp).cuda(), batch.offsets, nchunks, out, rets)
        torch.cuda.synchronize()
        r = rets.cpu().numpy()
        assert ((r == _lib.E_CORRUPT) | ((r >= 0) & (r <= chunk_len))).all(), (trial, r[:8])
        assert (out[nchunks * chunk_len:].cpu().numpy() == 0x5A).all(), trial


@pytest.mark.parametrize("codec,esz,ndims,chunk_len", [("delta", 1, 80, 1024), ("xff", 2, 8, 100), ("xff", 2, 128, 2040), ("delta", 1, 5, 77)])
def test_verbatim_batches_decode_and_reject_damaged_headers(sz, oracle, codec, esz, ndims, chunk_len):
    """batches whose chunks are too short for a stream group take verbatim_dense_kernel / verbatim_decode_kernel (api.hip):
    the oracle's streams decode to the data; a header that announces groups, another ndims, more samples than the chunk
    holds or than the stream carries is SPRINTZ_E_CORRUPT for that chunk only, and nothing is written past a chunk's range"""
    import torch
    rng = np.random.default_rng(chunk_len + ndims)
    nchunks = 257
    total = nchunks * chunk_len - chunk_len // 2
    data = gen_fuzz(rng, total, esz, 0)
    cd = sz.ChunkedCodec(codec, esz, ndims, chunk_len, device="cuda:0")
    streams = oracle.compress_chunks(codec, data, chunk_len, ndims)
    assert all(s_.size == 8 + min(chunk_len, total - c * chunk_len) * esz for c, s_ in enumerate(streams))     # verbatim indeed
    batch = cd.compress(torch.from_numpy(data).cuda())
    got, goffs, gsizes = batch.data.cpu().numpy(), batch.offsets.cpu().numpy(), batch.sizes.cpu().numpy()
    for c in range(nchunks):
        assert gsizes[c] == streams[c].size and np.array_equal(got[goffs[c]:goffs[c] + gsizes[c]], streams[c]), c
    damaged = [s_.copy() for s_ in streams]
    damaged[3][0] = 1                                                 # announces a group
    damaged[10][6] ^= 1                                               # another ndims
    damaged[20][4:6] = np.frombuffer(np.uint16(min(chunk_len + 1, 0xffff)).tobytes(), np.uint8)   # more samples than the chunk holds
    damaged[30] = damaged[30][:8 + (chunk_len // 2) * esz]           # the stream ends before its samples do
    offs = np.zeros(nchunks + 1, np.int64)
    offs[1:] = np.cumsum([s_.size for s_ in damaged])
    comp = torch.from_numpy(np.concatenate(damaged + [np.zeros(16, np.uint8)])).cuda()
    out = torch.full((nchunks * chunk_len,), 0x5a, dtype=cd.dtype, device="cuda:0")
    rets = torch.empty(nchunks, dtype=torch.int64, device="cuda:0")
    cd.decompress_into(comp, torch.from_numpy(offs).cuda(), nchunks, out, rets)
    o, r = out.cpu().numpy(), rets.cpu().numpy()
    fill = np.array([0x5a], dtype=np.uint8).view(np.int8)[0] if esz == 1 else 0x5a
    for c in range(nchunks):
        lo, n = c * chunk_len, min(chunk_len, total - c * chunk_len)
        if c in (3, 10, 20, 30):
            assert r[c] == -5, (c, r[c])                              # SPRINTZ_E_CORRUPT
            assert (o[lo:lo + chunk_len].view(np.uint8) == 0x5a).all() if esz == 1 else (o[lo:lo + chunk_len] == 0x5a).all(), c
        else:
            assert r[c] == n and np.array_equal(o[lo:lo + n].view(data.dtype), data[lo:lo + n]), c
            assert (o[lo + n:lo + chunk_len] == 0x5a).all(), c       # the rest of a short last chunk's range is not touched


def test_short_verbatim_chunks_leave_zero_padding(sz):
    """chunks too short for one group are stored verbatim (sprintz_xff_rle.cpp:116-124): the 16-byte alignment padding of the
    container must be zeros, not stale workspace bytes"""
    import torch
    rng = np.random.default_rng(5)
    nchunks, chunk_len, ndims = 200, 1021, 80                     # 1021 < 16 * 80: verbatim; 8 + 1021 = 1029 bytes -> 11 bytes of padding
    data = gen_fuzz(rng, nchunks * chunk_len, 1, 0)
    cd = sz.ChunkedCodec("delta", 1, ndims, chunk_len, device="cuda:0")
    cd.workspace(nchunks)["slots"].fill_(0xEE)                    # stale bytes in the slot buffer
    batch = cd.compress(torch.from_numpy(data).cuda())
    comp, offs, sizes = batch.data.cpu().numpy(), batch.offsets.cpu().numpy(), batch.sizes.cpu().numpy()
    for c in range(nchunks):
        assert sizes[c] == 8 + chunk_len
        assert not comp[offs[c] + sizes[c]:offs[c + 1]].any(), c


# ------------------------------------------------ optional Huffman stage (format: oracle/huf_oracle.c)

@pytest.mark.parametrize("step", [2, 8, 300])
def test_huffman_stage_matches_oracle_and_roundtrips(sz, oracle, step):
    """GPU Huffman records, tables and offsets are bit-exact with the CPU oracle of our container
    format; GPU decode rebuilds the exact Sprintz container; the data survives the whole chain."""
    import torch
    rng = np.random.default_rng(40 + step)
    codec, esz, ndims, chunk_len, nchunks = "xff", 2, 8, 5120, 150          # 3 segments, last one short
    data = np.concatenate([gen_walk(rng, 100 * chunk_len, ndims, esz, step, flat_every=5),
                           gen_fuzz(rng, 50 * chunk_len, esz, 0)])            # incompressible chunks -> stored records
    cd = sz.ChunkedCodec(codec, esz, ndims, chunk_len, device="cuda:0")
    batch = cd.compress(torch.from_numpy(data).cuda())
    hb = sz.huf_compress(batch)
    dense = batch.data.cpu().numpy()
    offs = batch.offsets.cpu().numpy().astype(np.uint64)
    sizes = batch.sizes.cpu().numpy().astype(np.uint32)
    want, want_offs, want_tables = oracle.huf_compress(dense, offs, sizes)
    got_offs = hb.offsets.cpu().numpy().astype(np.uint64)
    assert np.array_equal(got_offs, want_offs)
    assert np.array_equal(hb.tables.cpu().numpy(), want_tables)
    got = hb.data.cpu().numpy()
    assert np.array_equal(got[:want.size], want)                            # records, headers, zeroed gaps
    # cross-decode: the oracle understands the GPU's records
    od, oo, osz = oracle.huf_decompress(got[:want.size], got_offs, hb.tables.cpu().numpy(), int(offs[-1]))
    assert np.array_equal(osz, sizes) and np.array_equal(oo, offs)
    back = sz.huf_decompress(hb, int(offs[-1]))
    assert np.array_equal(back.offsets.cpu().numpy(), batch.offsets.cpu().numpy())
    assert np.array_equal(back.sizes.cpu().numpy(), batch.sizes.cpu().numpy())
    bd = back.data.cpu().numpy()
    for c in range(nchunks):
        assert np.array_equal(bd[int(offs[c]):int(offs[c]) + int(sizes[c])], dense[int(offs[c]):int(offs[c]) + int(sizes[c])]), c
    out = cd.decompress(back)
    assert np.array_equal(out.cpu().numpy(), data)
    if step <= 8:                                                            # a third of the chunks is incompressible noise
        assert hb.total_bytes() < 0.99 * batch.stream_bytes()


def test_huffman_odd_batches(sz, oracle):
    """chunk counts around the 64-chunk segment size, tiny and empty streams, byte-dense containers"""
    import torch
    rng = np.random.default_rng(77)
    for codec, esz, ndims, chunk_len, nchunks in (("delta", 1, 3, 40, 1), ("xff", 2, 8, 5120, 63), ("xff", 2, 8, 5120, 65),
                                                  ("delta", 1, 80, 1024, 129), ("xff", 1, 8, 16 * 8 + 5, 200)):
        data = gen_walk(rng, nchunks * chunk_len - chunk_len // 2, ndims, esz, 3, flat_every=3)
        cd = sz.ChunkedCodec(codec, esz, ndims, chunk_len, device="cuda:0")
        batch = cd.compress(torch.from_numpy(data).cuda())
        hb = sz.huf_compress(batch)
        dense, offs = batch.data.cpu().numpy(), batch.offsets.cpu().numpy().astype(np.uint64)
        sizes = batch.sizes.cpu().numpy().astype(np.uint32)
        want, want_offs, want_tables = oracle.huf_compress(dense, offs, sizes)
        assert np.array_equal(hb.offsets.cpu().numpy().astype(np.uint64), want_offs), (codec, nchunks)
        assert np.array_equal(hb.tables.cpu().numpy(), want_tables), (codec, nchunks)
        assert np.array_equal(hb.data.cpu().numpy()[:want.size], want), (codec, nchunks)
        rets = torch.empty(nchunks, dtype=torch.int64, device="cuda:0")
        back = sz.huf_decompress(hb, int(offs[-1]), rets=rets)
        assert np.array_equal(rets.cpu().numpy(), sizes.astype(np.int64))
        assert np.array_equal(cd.decompress(back).cpu().numpy(), data), (codec, nchunks)


def test_huffman_decoder_survives_damaged_containers(sz):
    """bit flips in records, headers, offsets and tables: no fault, no hang, no write outside the
    destination; chunks of untouched segments still decode exactly"""
    import torch
    rng = np.random.default_rng(99)
    codec, esz, ndims, chunk_len, nchunks = "xff", 2, 8, 5120, 256           # 4 segments
    data = gen_walk(rng, nchunks * chunk_len, ndims, esz, 6, flat_every=4)
    cd = sz.ChunkedCodec(codec, esz, ndims, chunk_len, device="cuda:0")
    batch = cd.compress(torch.from_numpy(data).cuda())
    hb = sz.huf_compress(batch)
    dense = batch.data.cpu().numpy()
    offs = batch.offsets.cpu().numpy()
    sizes = batch.sizes.cpu().numpy()
    cap = int(offs[-1])
    h_data, h_offs, h_tabs = hb.data.cpu().numpy().copy(), hb.offsets.cpu().numpy().copy(), hb.tables.cpu().numpy().copy()

    def run(d, o, t, check_segments):
        bad = sz.HufBatch(torch.from_numpy(d).cuda(), torch.from_numpy(o).cuda(), torch.from_numpy(t).cuda(), nchunks,
                          hb.total_len, hb.chunk_len, hb.ndims)
        rets = torch.empty(nchunks, dtype=torch.int64, device="cuda:0")
        back = sz.huf_decompress(bad, cap, rets=rets)
        torch.cuda.synchronize()
        guard = back.data[cap + 16 * nchunks:].cpu().numpy()
        assert (guard == 0).all()                                           # READ_SLACK past the capacity untouched
        bd, bo = back.data.cpu().numpy(), back.offsets.cpu().numpy()
        for c in check_segments:
            assert bo[c] == offs[c]
            assert np.array_equal(bd[int(offs[c]):int(offs[c]) + int(sizes[c])], dense[int(offs[c]):int(offs[c]) + int(sizes[c])]), c
        return rets.cpu().numpy()

    # (a) payload bit flips in segment 1 (chunks 64..127): sizes are unchanged, so every other segment is exact
    d = h_data.copy()
    lo, hi = int(h_offs[64]), int(h_offs[128])
    for pos in rng.integers(lo, hi, 200):
        if (pos - lo) % 4096 >= 12:                                         # leave most headers alone here
            d[pos] ^= 1 << int(rng.integers(0, 8))
    run(d, h_offs, h_tabs, list(range(0, 64)) + list(range(128, 256)))
    # (b) header damage: absurd symbol counts, sub-stream sizes, stored flags
    d = h_data.copy()
    for c in (3, 70, 200):
        o = int(h_offs[c])
        d[o:o + 4] = np.frombuffer(np.uint32(0x7fffffff).tobytes(), np.uint8)
    o = int(h_offs[10]); d[o + 4:o + 10] = 0xff
    o = int(h_offs[11]); d[o + 3] ^= 0x80
    r = run(d, h_offs, h_tabs, [])
    assert r[3] == sz._lib.E_CORRUPT and r[70] == sz._lib.E_CORRUPT and r[200] == sz._lib.E_CORRUPT and r[10] == sz._lib.E_CORRUPT
    # (c) offsets damage: not monotonic, misaligned, past the end
    o = h_offs.copy(); o[5] = o[9]; o[100] += 2; o[150] = o[-1] + 4096
    run(h_data, o, h_tabs, [])
    # (d) table damage: lengths that are no prefix code at all
    t = h_tabs.copy(); t[128:256] = 0xff; t[0:64] = 0
    run(h_data, h_offs, t, list(range(128, 256)))
    # and the device is still healthy
    r = run(h_data, h_offs, h_tabs, range(nchunks))
    assert np.array_equal(r, sizes.astype(np.int64))


@pytest.mark.parametrize("name,codec,esz,ndims,chunk_len", [c for c in CONFIGS if c[0] in ("cfg2", "cfg3_1k", "cfg3_10k", "cfg5", "xff8", "cfg1", "uni16_xff", "uni16_delta_ragged", "low8_d2", "low8_d4", "lowdim16", "low16_d2_delta_ragged", "lowdim8", "low8_d3_delta", "wide8_d128_xff", "wide8_d100_delta", "wide8_d66_xff", "wide16_d80_xff", "wide16_d128_delta", "wide16_d72_xff")])
def test_generic_kernels_agree_with_the_fast_ones(sz, request, name, codec, esz, ndims, chunk_len):
    """SPRINTZ_OPT_NO_FAST routes the same calls to decode_kernel.h / encode_kernel.h: same bytes, same samples"""
    import torch
    from sprintz_amd import _lib
    request.addfinalizer(lambda: _lib.set_option(_lib.OPT_NO_FAST, 0))
    rng = np.random.default_rng(zlib.crc32(name.encode()) + 1)
    data = np.concatenate([gen_walk(rng, 40 * chunk_len, ndims, esz, 5, flat_every=3), gen_fuzz(rng, 9 * chunk_len + 17 * ndims, esz, 2)])
    cd = sz.ChunkedCodec(codec, esz, ndims, chunk_len, device="cuda:0")
    x = torch.from_numpy(data).cuda()
    fast = cd.compress(x)
    _lib.check(_lib.set_option(_lib.OPT_ENC_PAIR, 0))          # one column per lane (encode_fast.h) where two are the default
    request.addfinalizer(lambda: _lib.set_option(_lib.OPT_ENC_PAIR, 1))
    fast1 = cd.compress(x)
    _lib.check(_lib.set_option(_lib.OPT_ENC_PAIR, 1))
    assert torch.equal(fast.sizes, fast1.sizes) and torch.equal(fast.data[: fast.total_bytes()], fast1.data[: fast1.total_bytes()])
    _lib.check(_lib.set_option(_lib.OPT_NO_FAST, 1))
    slow = cd.compress(x)
    assert torch.equal(fast.sizes, slow.sizes) and torch.equal(fast.offsets, slow.offsets)
    assert torch.equal(fast.data[: fast.total_bytes()], slow.data[: slow.total_bytes()])
    out_slow = cd.decompress(fast)
    _lib.check(_lib.set_option(_lib.OPT_NO_FAST, 0))
    out_fast = cd.decompress(slow)
    assert torch.equal(out_slow, out_fast) and np.array_equal(out_fast.cpu().numpy(), data)


def test_containers_beyond_4_GiB(sz):
    """520 000 chunks of mostly incompressible data: the container offsets pass 2^31 and 2^32, the output
    passes 2^32 bytes (a sign-extended 32-bit word in the decoder's wave base once broke this);
    also through the Huffman stage and the reduce-only query"""
    import torch
    n, chunk_len, ndims = 520000, 5120, 8
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    x = torch.randint(0, 65536, (n * chunk_len,), generator=g, device="cuda", dtype=torch.int32).to(torch.uint16)
    xv = x.view(torch.int16).view(n, chunk_len)
    xv[::4] = xv[::4] & 0x00ff                                   # every 4th chunk compresses 2:1 (coded Huffman records too)
    cd = sz.ChunkedCodec("xff", 2, ndims, chunk_len)
    batch = cd.compress(x)
    assert batch.total_bytes() > (1 << 32) + (1 << 28)
    rets = torch.empty(n, dtype=torch.int64, device="cuda")
    out = cd.decompress(batch, rets=rets)
    assert bool((rets == chunk_len).all()) and torch.equal(out, x)
    del out
    res, _ = cd.query(batch, "sum")
    want = x.view(torch.int16).to(torch.int64).bitwise_and(0xffff).view(-1, ndims).sum(dim=0)
    assert torch.equal(res, want)
    hb = sz.huf_compress(batch)
    back = sz.huf_decompress(hb, batch.total_bytes(), rets=rets)
    assert torch.equal(back.sizes, batch.sizes) and torch.equal(back.offsets, batch.offsets)
    out = cd.decompress(back)
    assert torch.equal(out, x)


@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("SPRINTZ_SWEEP", "200")))))
def test_random_shapes(sz, oracle, seed):
    """a sweep over the kernel variants' dispatch space: random element width, ndims, chunk length (multiples of
    16 and not), codec (the five), data mix -- batched compress bytes == oracle's, decompress == input"""
    import torch
    rng = np.random.default_rng(1000 + seed)
    esz = int(rng.choice([1, 2]))
    codec = str(rng.choice(["delta", "xff", "delta_norle", "bitpack", "xff_norle"]))
    if codec == "xff_norle":
        esz = 1
    ndims = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 16, 17, 31, 32, 33, 64, 65, 66, 80, 100, 127, 128, 129, 200]))
    rows = int(rng.integers(3, 400))
    chunk_len = rows * ndims + (int(rng.integers(0, ndims)) if rng.random() < 0.3 else 0)
    if rng.random() < 0.5:
        chunk_len = max(16, (chunk_len + 15) & ~15)
    chunk_len = min(max(chunk_len, 8), 40000)
    nchunks = int(rng.integers(3, 40))
    total = nchunks * chunk_len - int(rng.integers(0, chunk_len))
    top = 1 << (8 * esz)
    kind = int(rng.integers(0, 4))
    if kind == 0:
        data = gen_walk(rng, total, ndims, esz, int(rng.integers(1, 40)), flat_every=int(rng.integers(2, 6)))
    elif kind == 1:
        data = gen_fuzz(rng, total, esz, int(rng.integers(0, 4)))
    elif kind == 2:
        data = (np.cumsum(rng.integers(-3, 4, total)) % top).astype(DTYPES[esz])
        data[total // 3: total // 3 + 3 * chunk_len // 2] = data[total // 3]
    else:
        data = np.concatenate([gen_walk(rng, total // 2, ndims, esz, 300 if esz == 2 else 20, flat_every=3), gen_fuzz(rng, total - total // 2, esz, 1)])
    data = np.ascontiguousarray(data[:total])
    nchunks = (total + chunk_len - 1) // chunk_len
    cd = sz.ChunkedCodec(codec, esz, ndims, chunk_len, device="cuda:0")
    batch = cd.compress(torch.from_numpy(data.view(np.int8 if esz == 1 else np.int16)).cuda().view(cd.dtype))
    comp, offs, sizes = batch.data.cpu().numpy(), batch.offsets.cpu().numpy(), batch.sizes.cpu().numpy()
    if codec in ("delta", "xff"):
        want = oracle.compress_chunks(codec, data, chunk_len, ndims)
    else:
        raw = {"delta_norle": 0, "bitpack": 1, "xff_norle": 2}[codec]
        want = [oracle.compress_norle(raw, data[c * chunk_len:(c + 1) * chunk_len], ndims)[0] for c in range(nchunks)]
    tag = (seed, codec, esz, ndims, chunk_len)
    for c in range(nchunks):
        assert sizes[c] == want[c].size, (tag, c)
        assert np.array_equal(comp[offs[c]:offs[c] + sizes[c]], want[c]), (tag, c)
    out = cd.decompress(batch)
    assert np.array_equal(out.cpu().numpy().view(DTYPES[esz])[:total], data), tag


@pytest.mark.parametrize("codec,esz,ndims,chunk_len,nchunks", [
    ("xff", 2, 8, 5120, 1), ("xff", 2, 8, 5120, 31), ("xff", 2, 8, 5120, 33), ("xff", 2, 8, 5120, 20011), ("xff", 2, 8, 1280, 70001),
    ("delta", 1, 16, 4096, 777), ("xff", 2, 32, 5120, 1000), ("delta", 2, 64, 4096, 130), ("xff", 1, 5, 4000, 300),
    ("delta", 1, 80, 10240, 97), ("xff", 1, 72, 72 * 64, 1031),       # the split mapping's encoder (8 bits, 65 .. 80 columns)
    ("xff", 2, 80, 10240, 205), ("delta", 1, 128, 16384, 66),           # 64 lanes x two columns
    ("xff", 2, 7, 7 * 160, 999),                                         # an odd width at 16 bits: the last lane's pair is half genuine
    ("delta", 1, 1, 1024, 5000), ("xff", 1, 3, 3 * 400, 70001), ("xff", 2, 2, 2048, 513), ("delta", 2, 1, 1000, 255),   # low-dim: no dense tail in that encoder -- the entry point runs the two launches itself (a tail for 256 chunks a workgroup was built in round 5 and measured slower: 0.416 against 0.396 ms on BASELINE config 1)
    ("delta", 1, 80, 1024, 5003), ("xff", 2, 8, 100, 333), ("xff", 2, 128, 2000, 77), ("delta", 1, 5, 77, 1),   # chunks too short for a group: verbatim, written straight into the container
    ("delta", 1, 8, 128, 16384), ("delta", 1, 8, 128, 16385),          # either side of the one-workgroup size scan (the scan + copy side of the comparison)
    ("xff", 2, 4, 4000, 500), ("delta", 2, 3, 3000, 257),              # 3 and 4 uint16 columns: encode_fast.h on 4 lanes a chunk, the one shape of it that arms the tail
])
@pytest.mark.parametrize("enc_pair", [1, 0])
def test_container_built_inside_the_encode_launch(sz, oracle, request, codec, esz, ndims, chunk_len, nchunks, enc_pair):
    """sprintz_mi355x_compress_batch_dense (one launch: chained scan over workgroups + in-kernel copy, compact_tail.h) writes
    the container, offsets and sizes that compress_batch + compact(align 16) write -- and that the oracle specifies; with two
    columns per lane (encode_wide.h, the default for 5 .. 128 columns) and with one (encode_fast.h, SPRINTZ_OPT_ENC_PAIR 0)"""
    import torch
    from sprintz_amd import _lib
    _lib.check(_lib.set_option(_lib.OPT_ENC_PAIR, enc_pair))
    request.addfinalizer(lambda: _lib.set_option(_lib.OPT_ENC_PAIR, 1))
    rng = np.random.default_rng(nchunks * 7 + ndims)
    total = nchunks * chunk_len - (chunk_len // 3 if nchunks > 1 else 0)          # a short last chunk
    data = gen_walk(rng, total, ndims, esz, 8, flat_every=4)
    x = torch.from_numpy(data).cuda()
    cd = sz.ChunkedCodec(codec, esz, ndims, chunk_len, device="cuda:0")
    src = cd._padded_view(x)
    res = {}
    for mode in (0, 1):
        assert _lib.set_option(_lib.OPT_DENSE_MODE, mode) == 0
        try:
            cd._ws = {}
            dense = torch.full((nchunks * cd.slot_stride + 16,), 0xEE, dtype=torch.uint8, device="cuda")
            offs = torch.full((nchunks + 1,), -1, dtype=torch.int64, device="cuda")
            ws, dense, offs = cd.compress_dense(src, total, dense=dense, offsets=offs)
            torch.cuda.synchronize()
            res[mode] = (dense.cpu().numpy(), offs.cpu().numpy(), ws["sizes"].cpu().numpy().copy(), ws["rets"].cpu().numpy().copy())
        finally:
            _lib.set_option(_lib.OPT_DENSE_MODE, 1)
    d0, o0, s0, r0 = res[0]                              # mode 0: encode, then scan + copy -- round 2's path
    end = int(o0[-1])
    for mode in (1,):                                    # 1: the container built inside the encode launch
        d1, o1, s1, r1 = res[mode]
        assert np.array_equal(o0, o1) and np.array_equal(s0, s1) and np.array_equal(r0, r1), mode
        assert np.array_equal(d0[:end], d1[:end]), mode
        assert (d1[end:] == 0xEE).all(), mode                                     # nothing written past the container
    want = oracle.compress_chunks(codec, data, chunk_len, ndims)
    for c in range(0, nchunks, max(1, nchunks // 60)):
        assert s0[c] == want[c].size and np.array_equal(d0[o0[c]:o0[c] + s0[c]], want[c]), c
        assert o0[c] % 16 == 0 and (c == 0 or o0[c] == o0[c - 1] + ((s0[c - 1] + 15) & ~15))

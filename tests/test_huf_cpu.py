"""CPU tests of the optional Huffman stage's oracle (oracle/huf_oracle.c).

Parity status of this stage: UNPINNED against the reference (there is no Huffman
coder in dblalock/sprintz; the paper uses Huff0 from the author's lzbench fork).
What is checked: our container round-trips, the code lengths are a valid length-limited
prefix code no worse than 1 % from the order-0 entropy bound + table, and -- where the
system has libzstd (it exports Huff0's HUF_compress) -- that we compress the Sprintz
streams at least as well as Huff0 does per chunk (SURVEY.md 8c acceptance (iii))."""
import ctypes as C

import numpy as np
import pytest

from harness import gen_fuzz, gen_walk


def _container(oracle, data, chunk_len, ndims, codec="xff"):
    streams = oracle.compress_chunks(codec, data, chunk_len, ndims)
    sizes = np.array([s.size for s in streams], np.uint32)
    offs = np.zeros(len(streams) + 1, np.uint64)
    pos = 0
    for i, s in enumerate(streams):
        pos = (pos + 15) & ~15
        offs[i] = pos
        pos += s.size
    pos = (pos + 15) & ~15
    offs[-1] = pos
    dense = np.zeros(pos + 16, np.uint8)
    for i, s in enumerate(streams):
        dense[int(offs[i]):int(offs[i]) + s.size] = s
    return dense, offs, sizes, streams


def test_lengths_are_a_valid_limited_prefix_code(oracle):
    rng = np.random.default_rng(0)
    for trial in range(200):
        kind = trial % 5
        if kind == 0:
            counts = rng.integers(0, 1000, 256)
        elif kind == 1:
            counts = (rng.random(256) ** 8 * 1e6).astype(np.int64)         # very skewed -> depth limit bites
        elif kind == 2:
            counts = np.zeros(256, np.int64); counts[rng.integers(0, 256, rng.integers(1, 5))] = rng.integers(1, 100)
        elif kind == 3:
            counts = np.ones(256, np.int64)
        else:
            counts = np.array([int(1.6 ** min(i, 40)) for i in range(256)])   # Fibonacci-like: deep tree
        lens = oracle.huf_lengths(counts)
        nz = counts > 0
        assert (lens[~nz] == 0).all() and (lens[nz] >= 1).all() and lens.max() <= 11
        kraft = (2.0 ** -lens[nz].astype(float)).sum()
        assert kraft <= 1.0 + 1e-12
        if nz.sum() >= 2:
            assert kraft == 1.0                                            # slack is always given back
            p = counts[nz] / counts[nz].sum()
            entropy = -(p * np.log2(p)).sum()
            avg = (p * lens[nz]).sum()
            assert avg < entropy + 1.0


@pytest.mark.parametrize("kind,step", [("walk", 8), ("walk", 2), ("walk", 300), ("fuzz", 0)])
def test_container_roundtrip_and_gain(oracle, kind, step):
    rng = np.random.default_rng(3)
    nchunks, chunk_len, ndims = 150, 5120, 8                                 # 3 segments, the last one short
    data = gen_walk(rng, nchunks * chunk_len, ndims, 2, step, flat_every=5) if kind == "walk" else \
        gen_fuzz(rng, nchunks * chunk_len, 2, 0)
    dense, offs, sizes, streams = _container(oracle, data, chunk_len, ndims)
    huf, ho, tables = oracle.huf_compress(dense, offs, sizes)
    assert (ho[:-1] % 4 == 0).all() and ho[-1] == huf.size
    back, offs2, sizes2 = oracle.huf_decompress(huf, ho, tables, int(offs[-1]))
    assert np.array_equal(sizes2, sizes) and np.array_equal(offs2, offs)
    for i, s in enumerate(streams):
        assert np.array_equal(back[int(offs[i]):int(offs[i]) + s.size], s)
    total = int(sizes.astype(np.int64).sum())
    assert huf.size + tables.size <= total + 4 * nchunks + 4 * nchunks + tables.size   # never worse than stored
    if kind == "walk" and step <= 8:
        assert (huf.size + tables.size) / total < 0.95


def test_tiny_and_degenerate_chunks(oracle):
    """empty streams, 1-byte streams, single-symbol segments"""
    dense = np.zeros(64, np.uint8)
    dense[16:20] = 7
    offs = np.array([0, 16, 32, 48], np.uint64)
    sizes = np.array([0, 4, 1], np.uint32)
    dense[32] = 7
    huf, ho, tables = oracle.huf_compress(dense, offs, sizes)
    back, offs2, sizes2 = oracle.huf_decompress(huf, ho, tables, 64)
    assert np.array_equal(sizes2, sizes)
    assert (back[int(offs2[1]):int(offs2[1]) + 4] == 7).all() and back[int(offs2[2])] == 7


def test_at_least_as_good_as_system_huff0(oracle):
    try:
        z = C.CDLL("libzstd.so.1")
        z.HUF_compress.restype = C.c_size_t
        z.HUF_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        z.HUF_compressBound.restype = C.c_size_t
        z.HUF_compressBound.argtypes = [C.c_size_t]
        z.HUF_isError.restype = C.c_uint
        z.HUF_isError.argtypes = [C.c_size_t]
    except (OSError, AttributeError):
        pytest.skip("no libzstd with HUF_compress on this machine")
    rng = np.random.default_rng(1)
    for step in (2, 8, 40):
        data = gen_walk(rng, 128 * 5120, 8, 2, step)
        dense, offs, sizes, streams = _container(oracle, data, 5120, 8)
        huf, ho, tables = oracle.huf_compress(dense, offs, sizes)
        ours = huf.size + tables.size
        theirs = 0
        for s in streams:
            out = np.zeros(z.HUF_compressBound(s.size), np.uint8)
            r = z.HUF_compress(out.ctypes.data, out.size, s.ctypes.data, s.size)
            theirs += s.size if (r == 0 or z.HUF_isError(r)) else r
        assert ours <= theirs * 1.01, (step, ours, theirs)

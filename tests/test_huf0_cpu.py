"""Huff0 wire format, CPU side: the decoder restated in oracle/huf0_oracle.c against blocks
written by libzstd's HUF_compress -- committed ones (tests/golden/golden_huf0_v1.npz) and, where
the library exists, fresh ones, with its own HUF_decompress as the second opinion."""
import numpy as np
import pytest

from harness import Zstd, gen_walk


def test_oracle_decodes_the_committed_blocks(oracle, golden_huf0):
    manifest, arrays = golden_huf0
    assert len(manifest) >= 300
    kinds = set()
    for m in manifest:
        plain, blk = arrays["p%04d" % m["idx"]], arrays["b%04d" % m["idx"]]
        got, ret = oracle.huf0_decompress(blk, plain.size)
        assert ret == plain.size and np.array_equal(got, plain), m
        kinds.add(m["kind"])
    assert kinds == {"stored", "rle", "fse", "nibbles"}


def test_oracle_agrees_with_libzstd(oracle):
    try:
        z = Zstd()
    except (OSError, AttributeError):
        pytest.skip("no libzstd with the HUF_* exports on this machine")
    rng = np.random.default_rng(8)
    plains = []
    data = gen_walk(rng, 40 * 5120, 8, 2, 8, flat_every=3)
    plains += [np.ascontiguousarray(s) for s in oracle.compress_chunks("xff", data, 5120, 8)]
    for n in (12, 100, 3000, 20000):
        for k in (2, 7, 100, 256):
            p = 1.0 / np.arange(1, k + 1) ** 1.5
            plains.append(rng.choice(k, n, p=p / p.sum()).astype(np.uint8))
    coded = 0
    for s in plains:
        blk = z.huf_compress(s)
        theirs, r = z.huf_decompress(blk, s.size) if 1 < blk.size < s.size else (s, s.size)
        ours, ro = oracle.huf0_decompress(blk, s.size)
        assert r == ro == s.size and np.array_equal(ours, theirs) and np.array_equal(ours, s)
        coded += 1 < blk.size < s.size
    assert coded > 40


def test_oracle_rejects_damaged_blocks(oracle, golden_huf0):
    """truncations and bit flips either fail or decode to *something* of the right size -- never crash"""
    manifest, arrays = golden_huf0
    rng = np.random.default_rng(9)
    rejected = 0
    for m in [m for m in manifest if m["kind"] in ("fse", "nibbles")][:60]:
        plain, blk = arrays["p%04d" % m["idx"]], arrays["b%04d" % m["idx"]]
        for trial in range(6):
            bad = blk.copy()
            if trial < 3:
                bad[rng.integers(0, min(bad.size, 40))] ^= 1 << rng.integers(0, 8)
            else:
                bad = bad[: rng.integers(2, bad.size)]
            got, ret = oracle.huf0_decompress(bad, plain.size)
            assert ret == plain.size or ret < 0
            rejected += ret < 0
    assert rejected > 50


def _streams(oracle, rng):
    out = []
    for esz, D, codec in ((2, 8, "xff"), (1, 1, "delta"), (1, 80, "delta")):
        data = gen_walk(rng, 70 * 5120, D, esz, 8, flat_every=4)
        out += [np.ascontiguousarray(s) for s in oracle.compress_chunks(codec, data, 5120, D)]
    for n in (0, 1, 5, 11, 12, 13, 100, 1000, 5000, 70000):
        for k in (1, 2, 3, 17, 129, 256):
            p = 1.0 / np.arange(1, k + 1) ** 1.3
            out.append(rng.choice(k, n, p=p / p.sum()).astype(np.uint8))
    return out


def test_writer_blocks_are_read_by_libzstd(oracle):
    """what oracle_huf0_compress_batch (the writer's specification) emits is Huff0 to the library"""
    try:
        z = Zstd()
    except (OSError, AttributeError):
        pytest.skip("no libzstd with the HUF_* exports on this machine")
    rng = np.random.default_rng(21)
    streams = _streams(oracle, rng)
    sizes = np.array([s.size for s in streams], np.uint32)
    offs = np.zeros(len(streams) + 1, np.uint64)
    offs[1:] = np.cumsum(sizes)
    blocks, bo = oracle.huf0_compress(np.concatenate(streams + [np.zeros(8, np.uint8)]), offs, sizes)
    coded = 0
    for c, s in enumerate(streams):
        blk = blocks[int(bo[c]):int(bo[c + 1])]
        assert blk.size <= s.size
        if 1 < blk.size < s.size:
            back, r = z.huf_decompress(blk, s.size)
            assert r == s.size and np.array_equal(back, s), c
            coded += 1
        ours, ro = oracle.huf0_decompress(blk, s.size) if s.size else (s, 0)
        assert ro == s.size and np.array_equal(ours, s), c
    assert coded > 150
    assert blocks.size < 0.97 * sizes.sum()

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

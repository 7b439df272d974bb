"""Non-RLE codecs (sprintz_delta.cpp:64-1391), CPU side: the oracle against streams minted
from the compiled reference, and against the reference itself where it was built."""
import numpy as np

from harness import DTYPES


def test_oracle_matches_reference_streams(oracle, golden_norle):
    manifest, arrays = golden_norle
    assert len(manifest) >= 1400
    for m in manifest:
        x, stream = arrays[m["name"] + "_in"], arrays[m["name"] + "_stream"]
        got, ret = oracle.compress_norle(m["raw"], x, m["ndims"])
        assert ret == m["ret"] and np.array_equal(got, stream), m
        back, n = oracle.decompress_norle(m["raw"], stream, m["esz"])
        assert n == m["n"] and np.array_equal(back, x), m


def test_oracle_vs_compiled_reference(oracle, reference):
    import pytest
    if not reference.has_norle():
        pytest.skip("oracle/_ref built before the non-RLE shim was added")
    rng = np.random.default_rng(8)
    for esz in (1, 2):
        for raw in (0, 1):
            for D in (1, 6, 9, 31, 64, 100):
                for n in (5, 200, 16 * D + 9, 3001):
                    x = (np.cumsum(rng.integers(-6, 7, n)) % (1 << (8 * esz))).astype(DTYPES[esz])
                    so, ro = oracle.compress_norle(raw, x, D)
                    buf, rr = reference.compress_norle_raw(raw, x, D)
                    assert ro == rr and np.array_equal(buf[:len(so)], so), (esz, raw, D, n)
                    back, _ = reference.decompress_norle(raw, so, esz)
                    assert np.array_equal(back, x)

"""Query-on-compressed, CPU side: the oracle's restatement of the reference's
*_rowmajor_*_rle_* family against streams minted from the compiled reference, and the
semantic definition of the reductions (oracle_query) against numpy."""
import numpy as np
import pytest

from harness import DTYPES


def _col_reduce(x, D, op):
    res = np.zeros(D, np.uint64)
    for c in range(D):
        col = x[c::D].astype(np.uint64)
        if col.size:
            res[c] = col.max() if op == 1 else col.sum()
    return res


def test_oracle_rowmajor_family_matches_reference_streams(oracle, golden_rowmajor):
    manifest, arrays = golden_rowmajor
    assert len(manifest) >= 800
    for m in manifest:
        x, stream = arrays[m["name"] + "_in"], arrays[m["name"] + "_stream"]
        so, ret = oracle.compress_rowmajor(m["codec"], x, m["ndims"])
        assert ret == m["ret"], m
        assert np.array_equal(so, stream), m


@pytest.mark.parametrize("op", [1, 2])
def test_oracle_query_is_the_reduction_of_the_decompressed_data(oracle, golden_rowmajor, op):
    manifest, arrays = golden_rowmajor
    for m in manifest[::3]:
        x, stream = arrays[m["name"] + "_in"], arrays[m["name"] + "_stream"]
        data, res = oracle.query(m["codec"], stream, m["esz"], m["n"], op, general=True)
        assert np.array_equal(data, x), m                       # lossless, also where the reference's own query is not
        assert np.array_equal(res, _col_reduce(x, m["ndims"], op)), m


def test_reference_query_materialize_where_it_works(reference, golden_rowmajor):
    """documents the upstream state: query_rowmajor_*(materialize) reproduces the data for
    delta 8/16 and xff 8 (test/test_query.cpp:59-120,180-200), not for xff 16 (:218-241)"""
    if not reference.has_query():
        pytest.skip("oracle/_ref built before the query shim was added")
    manifest, arrays = golden_rowmajor
    for m in manifest[::7]:
        x, stream = arrays[m["name"] + "_in"], arrays[m["name"] + "_stream"]
        d, ret = reference.query(m["codec"], stream, m["esz"], m["n"], 0, True, m["ndims"])
        ok = ret == m["n"] and np.array_equal(d, x)
        assert ok == m["ref_query_materialize_ok"], m
        if (m["codec"], m["esz"]) != ("xff", 2):
            assert ok, m


def test_sprintz_h_layout_query(oracle):
    """query over streams of the sprintz.h entry points (low-dim layout for small ndims)"""
    rng = np.random.default_rng(5)
    for esz in (1, 2):
        for codec in ("delta", "xff"):
            for D in (1, 2, 4, 8):
                x = (np.cumsum(rng.integers(-2, 3, 999)) % (1 << (8 * esz))).astype(DTYPES[esz])
                stream, _ = oracle.compress(codec, x, D)
                for op in (1, 2):
                    data, res = oracle.query(codec, stream, esz, x.size, op, general=False)
                    assert np.array_equal(data, x)
                    assert np.array_equal(res, _col_reduce(x, D, op))

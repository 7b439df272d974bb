"""CPU tests: oracle vs the compiled reference itself (oracle/_ref), over the
input families and sizes of the reference's own test-suite
(cpp/Compress/test/compress_testing.hpp:124-204,452-486).  Skipped when the
reference has not been built (it can only be built where /root/reference is)."""
import numpy as np
import pytest

from harness import (REF_TEST_SIZES, gen_fuzz, gen_patterns, gen_sparse, gen_walk)


def _check(oracle, reference, codec, data, ndims):
    esz = data.dtype.itemsize
    so, ro = oracle.compress(codec, data, ndims)
    buf, rr = reference.compress_raw(codec, data, ndims)
    assert ro == rr
    assert np.array_equal(buf[:so.size], so)
    do, dro = oracle.decompress(codec, so, esz, data.size)
    assert dro == data.size and np.array_equal(do, data.ravel())
    dr, drr = reference.decompress(codec, so, esz, data.size, ndims)
    if not (drr == data.size and np.array_equal(dr, data.ravel())):
        # only legal divergence: the reference's 16-bit FIRE run replay
        assert codec == "xff" and esz == 2 and ndims >= 3
        dq, _ = oracle.decompress(codec, so, esz, data.size, quirk=1)
        assert np.array_equal(dq, dr)


@pytest.mark.parametrize("esz", [1, 2])
@pytest.mark.parametrize("codec", ["delta", "xff"])
def test_families_match_reference(oracle, reference, codec, esz):
    rng = np.random.default_rng(123)
    for ndims in [1, 2, 3, 4, 5, 7, 8, 9, 16, 17, 31, 33, 64, 65, 80, 129]:
        for n in REF_TEST_SIZES + [5120]:
            for _, d in gen_patterns(n, esz):
                _check(oracle, reference, codec, d, ndims)
            for sh in (0, 2, 4, 6, 9, 12, 14):
                if sh < 8 * esz:
                    _check(oracle, reference, codec, gen_fuzz(rng, n, esz, sh), ndims)
            _check(oracle, reference, codec, gen_sparse(rng, n, esz, 0.02), ndims)
            _check(oracle, reference, codec, gen_walk(rng, n, ndims, esz, 8), ndims)
            _check(oracle, reference, codec, gen_walk(rng, n, ndims, esz, 30, flat_every=3), ndims)


@pytest.mark.parametrize("esz", [1, 2])
@pytest.mark.parametrize("codec", ["delta", "xff"])
def test_write_size_false_matches_reference(oracle, reference, codec, esz):
    rng = np.random.default_rng(5)
    for ndims in (1, 3, 8, 17):
        for n in (100, 16 * ndims * 5 + 3):
            d = gen_walk(rng, n, ndims, esz, 8)
            so, ro = oracle.compress(codec, d, ndims, write_size=False)
            buf, rr = reference.compress_raw(codec, d, ndims, write_size=False)
            assert ro == rr and np.array_equal(buf[:so.size], so)


def test_long_runs_and_run_cap(oracle, reference):
    """runs > 127 blocks (2-byte varint) and > 32767 blocks (cap, sprintz_xff_rle.cpp:71,455)"""
    for esz, codec, nd in [(1, "delta", 5), (2, "xff", 8), (1, "xff", 2), (2, "delta", 1)]:
        for nblocks in (130, 32767 + 5, 70000):
            n = nblocks * 8 * nd + 3
            d = np.zeros(n, np.uint8 if esz == 1 else np.uint16)
            d[-2:] = 7
            _check(oracle, reference, codec, d, nd)


def test_big_stream(oracle, reference):
    """1024*1024+7 elements, as compress_testing.hpp:462"""
    rng = np.random.default_rng(9)
    n = 1024 * 1024 + 7
    for esz, codec, nd in [(1, "xff", 8), (2, "xff", 8), (2, "delta", 32), (1, "delta", 1)]:
        _check(oracle, reference, codec, gen_walk(rng, n, nd, esz, 8, flat_every=5), nd)

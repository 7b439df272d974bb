"""Known answers for the codec's primitives through the C-ABI (-m gpu): for EVERY value of an 8-bit and a
16-bit error the HIP encoders write the closed-form header field and payload (needed_nbits, zigzag, the 7 -> 8 / 15 -> 16 rounding
per byte: what cpp/Compress/test/test_bitpack.cpp:55-240 and test/test_sprintz_delta.cpp:24-68 pin with tables), byte for byte the
oracle's streams, and the HIP decoders invert every one of them; and the FIRE forecaster's counters are driven through every
truncated coefficient (sprintz_xff_rle.cpp:217) and far into both signs of the untruncated one (sprintz_xff_lowdim.cpp:170-173).
Both kernel families (decode_path: one workgroup per chunk / one lane per column); batched and single-call entry points."""
import numpy as np
import pytest

import kat
from test_primitives_cpu import SHAPES, check_kat_streams

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("decode_path")]


@pytest.fixture(scope="module")
def sz():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import sprintz_amd
    return sprintz_amd


def split(comp, offs):
    return [comp[int(offs[c]):int(offs[c + 1])] for c in range(len(offs) - 1)]


@pytest.mark.parametrize("w,ndims,lowdim", SHAPES)
def test_every_error_value_through_the_batched_entry_points(sz, oracle, w, ndims, lowdim):
    x = kat.kat_chunks(w, ndims)
    chunk_len = kat.kat_rows(ndims) * ndims
    comp, offs = sz.compress_chunked("delta", x.ravel(), ndims, chunk_len)
    streams = split(comp, offs)
    assert len(streams) == 1 << w
    check_kat_streams(streams, w, ndims, lowdim)                       # the closed forms
    want = oracle.compress_chunks("delta", x.ravel(), chunk_len, ndims)
    assert all(np.array_equal(a, b) for a, b in zip(streams, want))    # and the oracle's bytes, every chunk
    dec = sz.decompress_chunked("delta", comp, offs, w // 8, ndims, chunk_len)
    assert np.array_equal(dec, x.ravel())
    # the FIRE codecs on the same chunks: block 0 is coded with coefficient 0 (the same fields), the later blocks are not runs any more
    comp, offs = sz.compress_chunked("xff", x.ravel(), ndims, chunk_len)
    want = oracle.compress_chunks("xff", x.ravel(), chunk_len, ndims)
    assert all(np.array_equal(a, b) for a, b in zip(split(comp, offs), want))
    fields, _, _ = kat.expected_fields(w, ndims, lowdim)
    for e in range(0, 1 << w, 251):
        assert np.array_equal(kat.read_fields(want[e], w, ndims)[:ndims], fields[e]), e
    assert np.array_equal(sz.decompress_chunked("xff", comp, offs, w // 8, ndims, chunk_len), x.ravel())


@pytest.mark.parametrize("w,ndims,lowdim", [(8, 8, False), (8, 1, True), (16, 8, False), (16, 1, True)])
def test_every_8_bit_and_a_stride_of_16_bit_error_values_single_call(sz, oracle, w, ndims, lowdim):
    """the drop-in symbols on the same chunks (every value at 8 bits; every 61st at 16 bits plus the width boundaries 2^k - 1, 2^k)"""
    x = kat.kat_chunks(w, ndims)
    n = kat.kat_rows(ndims) * ndims
    vals = set(range(256)) if w == 8 else set(range(0, 65536, 61))
    for k in range(w + 1):
        for v in ((1 << k) - 1, 1 << k, (1 << w) - (1 << k), (1 << w) - (1 << k) - 1 if k else 0):
            vals.add(v & ((1 << w) - 1))
    comp_fn = getattr(sz, f"sprintz_compress_delta_{w}b")
    dec_fn = getattr(sz, f"sprintz_decompress_delta_{w}b")
    streams = {}
    for e in sorted(vals):
        data = x[e].ravel()
        dest = np.full(n * 3 + 256, 0xAB, np.uint8)
        ret = comp_fn(data, n, dest, ndims, True)
        want, wret = oracle.compress("delta", data, ndims)
        assert ret == wret and np.array_equal(dest[:want.size], want) and (dest[want.size:] == 0xAB).all(), (e, sz.last_error())
        out = np.zeros(n + 64, data.dtype)
        assert dec_fn(want, out) == n and np.array_equal(out[:n], data) and not out[n:].any(), e
        streams[e] = want
    full = [streams.get(e) for e in range(1 << w)]
    if w == 8:
        check_kat_streams(full, w, ndims, lowdim)


@pytest.mark.parametrize("w,ndims,lowdim", [(16, 8, False), (8, 8, False), (16, 1, True), (16, 2, True), (8, 1, True), (8, 4, True), (16, 5, False), (8, 24, False)])
def test_fire_coefficient_boundaries(sz, oracle, w, ndims, lowdim):
    """the counters driven through every truncated coefficient of the general layout and far into both signs of the untruncated
    low-dim one (test_primitives_cpu.py says these inputs get there): the oracle's bytes, the input back"""
    nblocks = 40
    x = kat.fire_boundary_chunks(w, ndims, 3000, nblocks, seed=w * 100 + ndims)
    chunk_len = 8 * nblocks * ndims
    comp, offs = sz.compress_chunked("xff", x.ravel(), ndims, chunk_len)
    want = oracle.compress_chunks("xff", x.ravel(), chunk_len, ndims)
    got = split(comp, offs)
    bad = [c for c in range(len(want)) if not np.array_equal(got[c], want[c])]
    assert not bad, bad[:8]
    assert np.array_equal(sz.decompress_chunked("xff", comp, offs, w // 8, ndims, chunk_len), x.ravel())

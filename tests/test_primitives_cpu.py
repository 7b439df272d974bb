"""Known answers for the codec's primitives, CPU leg: the oracle against closed forms for EVERY value of
an 8-bit and a 16-bit error -- what cpp/Compress/test/test_bitpack.cpp:55-240 and test/test_sprintz_delta.cpp:24-68 pin with tables
(needed_nbits, zigzag, the 7 -> 8 / 15 -> 16 rounding per byte, the field <-> width mapping), here pinned through the stream bytes.
The GPU leg (test_gpu_primitives.py) pushes the same chunks through the C-ABI."""
import numpy as np
import pytest

import kat

# (width, columns, low-dim layout?)
SHAPES = [(8, 8, False), (8, 5, False), (8, 1, True), (8, 3, True), (8, 4, True),
          (16, 8, False), (16, 3, False), (16, 1, True), (16, 2, True)]


def check_kat_streams(streams, w, ndims, lowdim):
    """streams[e]: np.uint8 stream of chunk e (kat.kat_chunks); every header field, every payload byte, the run and the tail"""
    fields, nb, z = kat.expected_fields(w, ndims, lowdim)
    err = kat.kat_errors(w, ndims)
    hb = 3 if w == 8 else 4
    hdr = (2 * ndims * hb + 7) // 8
    esz = w // 8
    nblocks = kat.kat_rows(ndims) // 8
    # the delta codecs close a run when fewer than two blocks would remain (`<`, sprintz_delta_rle.cpp:226): blocks 1 .. nblocks - 3 are
    # the run, the last two blocks travel verbatim
    run, tail_rows = (1, 0) if nblocks == 2 else (nblocks - 3, 16)
    for e in range(1 << w):
        s = streams[e]
        if not fields[e].any():
            continue                                                 # (chunk 0 of a one-column shape: all zero, nothing of block 0 to read)
        assert int.from_bytes(bytes(s[0:4]), "little") == 1, (e, "one group")
        assert int(s[4]) | int(s[5]) << 8 == tail_rows * ndims and int(s[6]) | int(s[7]) << 8 == ndims, e
        got = kat.read_fields(s, w, ndims)
        assert np.array_equal(got[:ndims], fields[e]), (w, ndims, lowdim, e, got[:ndims], fields[e])
        assert not got[ndims:].any(), (e, "slot 1 is a run")
        payload = kat.expected_block_payload(z[e], nb[e], w, lowdim)
        at = 8 + hdr
        assert bytes(s[at:at + len(payload)]) == payload, (w, ndims, lowdim, e)
        at += len(payload)
        assert s[at] == run, (e, s[at], run)
        tail = np.tile(err[e], tail_rows).astype(np.uint8 if w == 8 else np.uint16).tobytes()
        assert bytes(s[at + 1:]) == tail, (w, ndims, lowdim, e)


@pytest.mark.parametrize("w,ndims,lowdim", SHAPES)
def test_oracle_matches_closed_form_for_every_error_value(oracle, w, ndims, lowdim):
    x = kat.kat_chunks(w, ndims)
    streams = oracle.compress_chunks("delta", x.ravel(), kat.kat_rows(ndims) * ndims, ndims)
    assert len(streams) == 1 << w
    check_kat_streams(streams, w, ndims, lowdim)
    # and back: the oracle's decoder inverts every one of them (zigzag^-1 and the field -> width mapping for every value)
    comp = np.concatenate(streams)
    offs = np.zeros(len(streams) + 1, np.uint64)
    offs[1:] = np.cumsum([s.size for s in streams])
    dec = oracle.decompress_chunks("delta", comp, offs[:-1], w // 8, kat.kat_rows(ndims) * ndims, x.size)
    assert np.array_equal(dec, x.ravel())


def test_zigzag_and_widths_closed_form_tables():
    """the closed forms themselves against first principles, every value (test_bitpack.cpp's tables restated)"""
    for w in (8, 16):
        e = np.arange(1 << w)
        s = np.where(e >> (w - 1), e - (1 << w), e)
        z = kat.zigzag(e, w)
        assert np.array_equal(z, np.where(s >= 0, 2 * s, -2 * s - 1))
        assert sorted(z.tolist()) == list(range(1 << w))             # a bijection
        nb = kat.nbits_general(z, w)
        assert set(np.unique(nb).tolist()) == ({0, 1, 2, 3, 4, 5, 6, 8} if w == 8 else {0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 11, 12, 13, 14, 16})
        assert (z < (1 << nb.astype(np.int64))).all()                # the value fits its width
        nl = kat.nbits_lowdim(z, w)
        assert set(np.unique(nl).tolist()) == set(range(w - 1)) | {w}
        assert (kat.field_of(nb, w) < (8 if w == 8 else 16)).all()


@pytest.mark.parametrize("w,ndims,lowdim", [(16, 8, False), (8, 8, False), (16, 1, True), (16, 2, True), (8, 1, True), (8, 4, True)])
def test_fire_boundary_inputs_visit_the_coefficient_range(oracle, w, ndims, lowdim):
    """the inputs of the FIRE boundary test (GPU leg) do what they are for: the counters travel through every truncated coefficient
    value of the general layout (sprintz_xff_rle.cpp:217: 16 values of the top nibble at 16 bits, every multiple of 16 within what 40 blocks
    reach at 8), and far into both signs in the low-dim layouts (sprintz_xff_lowdim.cpp:170-173); the model that says so
    agrees with the oracle on the widths of every first group"""
    nblocks = 40
    x = kat.fire_boundary_chunks(w, ndims, 3000, nblocks, seed=w * 100 + ndims)
    coefs = np.stack(kat.fire_coefficients(x, w, lowdim))            # [blocks, chunks, columns]
    if not lowdim and w == 16:
        assert set(np.unique((coefs >> 12) & 15).tolist()) == set(range(16))
    elif not lowdim:
        seen = set(np.unique(coefs).tolist())
        # (an 8-bit column's gradient moves its counter by at most 32 a block: +-512 is what 40 blocks reach; every multiple of 16 between)
        assert set(range(-512, 513, 16)) <= seen, sorted(seen)[:4]
    elif w == 16:
        assert coefs.min() < -(1 << 16) and coefs.max() > (1 << 16), (coefs.min(), coefs.max())   # beyond 2^16 the 32-bit product with a 16-bit delta wraps (quirk 4)
    else:
        assert coefs.min() <= -512 and coefs.max() >= 512 and len(np.unique(coefs)) > 900, (coefs.min(), coefs.max())   # untruncated: every value between
    # the model against the oracle: block 1's widths depend on block 0's counters
    streams = oracle.compress_chunks("xff", x.ravel(), 8 * nblocks * ndims, ndims)
    hb = 3 if w == 8 else 4
    mism = 0
    for c in range(0, 3000, 7):
        got = kat.read_fields(streams[c], w, ndims)
        xc = x[c].astype(np.int64)
        co = coefs[:2, c, :]
        want = []
        pv, pd = np.zeros(ndims, np.int64), np.zeros(ndims, np.int64)
        for b in range(2):
            mask = np.zeros(ndims, np.int64)
            for i in range(8):
                delta = kat._wrap(xc[8 * b + i] - pv, w)
                pred = kat._wrap(kat._wrap(pd * co[b], 32) >> w, w)
                mask |= kat.zigzag(kat._wrap(delta - pred, w) & ((1 << w) - 1), w)
                pv, pd = xc[8 * b + i], delta
            nb = kat.nbits_lowdim(mask, w) if lowdim else kat.nbits_general(mask, w)
            want.append(kat.field_of(nb, w))
        want = np.concatenate(want)
        if want[:ndims].any() and want[ndims:].any():                # (no run in the first group: its fields are the two blocks')
            mism += int(not np.array_equal(got, want))
    assert mism == 0

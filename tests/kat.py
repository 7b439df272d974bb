"""Closed-form known answers for the codec's primitives, and the inputs that pin them -- shared by the CPU leg (oracle ==
closed form) and the GPU leg (HIP path through the C-ABI == oracle == closed form).

What the reference pins with exhaustive tables (cpp/Compress/test/test_bitpack.cpp:55-240: `needed_nbits_*` / zigzag for EVERY
int8 / int16 value; test/test_sprintz_delta.cpp:24-68: the nbits -> mask table) is pinned here through the codec itself: for every
value e of a w-bit error there is one 16-row chunk whose first block has exactly that error in column 0 (delta codec: rows 0 and
1 differ by e), so the stream's header field must be the closed-form width of zigzag(e) and its payload the closed-form packing.

Nothing here is product code or oracle code: plain numpy restatements of SURVEY.md Appendix A.2-A.4 (bitpack.h:72-93,
302-358; sprintz_xff_rle.cpp:259-265,296; sprintz_delta_lowdim.cpp:173).
"""
import numpy as np


def zigzag(e, w):
    """bitpack.h:302-358: (x << 1) ^ (x >> (w - 1)) on w-bit two's complement; e given as unsigned w-bit values"""
    e = np.asarray(e, np.int64) & ((1 << w) - 1)
    s = np.where(e >> (w - 1), e - (1 << w), e)                 # signed value
    return ((s << 1) ^ (s >> (w - 1))) & ((1 << w) - 1)


def bitlen(v):
    v = np.asarray(v, np.int64)
    out = np.zeros(v.shape, np.int64)
    for b in range(17):
        out = np.where(v >> b, b + 1, out)
    return out


def nbits_general(z, w):
    """general layout (A.3): w = 8: bit length with 7 -> 8; w = 16: per BYTE -- a non-zero high byte gives 8 + its bit length with
    7 -> 8 (i.e. 15 -> 16), otherwise the low byte's bit length with 7 -> 8 (sprintz_xff_rle.cpp:259-265, bitpack.h:72-93)"""
    z = np.asarray(z, np.int64)
    r8 = lambda n: np.where(n == 7, 8, n)                        # noqa: E731
    if w == 8:
        return r8(bitlen(z))
    hi, lo = z >> 8, z & 0xff
    return np.where(hi != 0, 8 + r8(bitlen(hi)), r8(bitlen(lo)))


def nbits_lowdim(z, w):
    """low-dim layouts (A.3, last line): only (w - 1) -> w (sprintz_delta_lowdim.cpp:173, sprintz_xff_lowdim.cpp:207-208)"""
    n = bitlen(z)
    return np.where(n == w - 1, w, n)


def field_of(nbits, w):
    """header field of a width: the width itself below w - 1, w - 1 for w (A.2; :296, decode :747-749)"""
    return np.where(nbits == w, w - 1, nbits)


def kat_errors(w, ndims):
    """errors[e, d] (unsigned w-bit) of chunk e: column 0 carries e itself -- every value once -- the other columns other values of
    other widths, so that e's field sits at every bit offset a row can give it"""
    e = np.arange(1 << w, dtype=np.int64)[:, None]
    d = np.arange(ndims, dtype=np.int64)[None, :]
    return (e * (2 * d + 1) + 37 * d * d) & ((1 << w) - 1)


def kat_rows(ndims):
    """whole groups, and at least the 128 elements below which the reference stores a chunk verbatim (sprintz_xff_rle.cpp:116-124)"""
    return 16 * -(-128 // (16 * ndims))


def kat_chunks(w, ndims):
    """(2^w chunks) x kat_rows x ndims: row 0 zero, every later row = the chunk's errors (the delta codec's first block then has
    the deltas 0, e, 0, 0, 0, 0, 0, 0 in every column; every later block is all zero: one run, then the encoder's two-block tail)"""
    err = kat_errors(w, ndims)
    x = np.zeros((1 << w, kat_rows(ndims), ndims), np.int64)
    x[:, 1:, :] = err[:, None, :]
    return x.astype(np.uint8 if w == 8 else np.uint16)


def expected_fields(w, ndims, lowdim):
    """closed-form header fields of block 0, per chunk and column"""
    z = zigzag(kat_errors(w, ndims), w)
    nb = nbits_lowdim(z, w) if lowdim else nbits_general(z, w)
    return field_of(nb, w), nb, z


def read_fields(stream, w, ndims):
    """the 2 * ndims header fields of a stream's first group (A.2: LSB-first bit stream of hb-bit fields behind the 8-byte header)"""
    hb = 3 if w == 8 else 4
    nbytes = (2 * ndims * hb + 7) // 8
    bits = np.unpackbits(np.asarray(stream[8:8 + nbytes], np.uint8), bitorder="little")
    return (bits[: 2 * ndims * hb].reshape(2 * ndims, hb) * (1 << np.arange(hb))).sum(axis=1)


def expected_block_payload(z_row1, nb, w, lowdim):
    """the 8 rows of block 0 as the format packs them, for ONE chunk: row 1 holds the zigzagged errors, the other rows zero.
    general: 8 rows of ceil(sum(nb) / 8) bytes, fields LSB-first in column order (A.2); low-dim: per column nb bytes holding the
    column's 8 values of nb bits each, sample 0 first (A.4)"""
    z_row1 = [int(v) for v in z_row1]
    nb = [int(v) for v in nb]
    if lowdim:
        out = bytearray()
        for zc, n in zip(z_row1, nb):
            acc = zc << n                                        # sample 1 sits n bits in
            out += acc.to_bytes(n, "little") if n else b""
        return bytes(out)
    rowbytes = (sum(nb) + 7) // 8
    acc, at = 0, 0
    for zc, n in zip(z_row1, nb):
        acc |= (zc & ((1 << n) - 1)) << at
        at += n
    return bytes(rowbytes) + acc.to_bytes(rowbytes, "little") + bytes(6 * rowbytes)


# ---------------------------------------------------------------------------------------------------------------------------
# The FIRE forecaster's counters (A.3): a numpy model used ONLY to say which coefficients a set of inputs visits (coverage of the
# truncation boundaries of sprintz_xff_rle.cpp:217 / the untruncated sprintz_xff_lowdim.cpp:170-173), never as the expected output
# -- that is the oracle's.

def _wrap(v, bits):
    v = np.asarray(v, np.int64) & ((1 << bits) - 1)
    return np.where(v >> (bits - 1), v - (1 << bits), v)


def fire_coefficients(x, w, lowdim):
    """x: [chunks, rows, ndims] unsigned; -> list over blocks of the coefficient each (chunk, column) used for that block"""
    x = np.asarray(x, np.int64)
    nch, rows, nd = x.shape
    cbits = 16 if w == 8 else 32
    prev_val = np.zeros((nch, nd), np.int64)
    prev_delta = np.zeros((nch, nd), np.int64)
    counter = np.zeros((nch, nd), np.int64)
    coefs = []
    for b in range(rows // 8):
        if lowdim:
            coef = counter >> 1
        else:
            coef = _wrap((counter >> (1 + (w - 4))) << (w - 4), 16)
        coefs.append(coef.copy())
        grad = np.zeros((nch, nd), np.int64)
        for i in range(8):
            xi = x[:, 8 * b + i, :]
            delta = _wrap(xi - prev_val, w)
            pred = _wrap(_wrap(prev_delta * coef, 32) >> w, w)     # 32-bit wrapping product at both widths (sprintz_xff_lowdim.cpp:183)
            err = _wrap(delta - pred, w)
            if i & 1:
                sgn = np.sign(err)
                grad = _wrap(grad + _wrap(sgn * prev_delta, w), w)
            prev_val = xi
            prev_delta = delta
        counter = _wrap(counter + (grad >> 2), cbits)
    return coefs


def fire_boundary_chunks(w, ndims, nchunks, nblocks, seed):
    """inputs that drive the counters far in both directions: per (chunk, column) the delta alternates between two magnitudes with a
    period and a growth of its own -- accelerating where the forecast lags (errors and deltas of one sign: the counter climbs),
    oscillating where it overshoots (opposite signs: it falls), with magnitudes up to the width's range"""
    rng = np.random.default_rng(seed)
    rows = 8 * nblocks
    amp = np.exp(rng.uniform(np.log(2.0), np.log(float(1 << (w - 2))), size=(nchunks, 1, ndims)))
    period = rng.integers(1, 9, size=(nchunks, 1, ndims))
    growth = rng.uniform(-0.3, 1.5, size=(nchunks, 1, ndims))
    t = np.arange(rows)[None, :, None]
    sign = np.where(((t // period) & 1) == 1, -1.0, 1.0) * np.where(rng.random((nchunks, 1, ndims)) < 0.5, 1.0, -1.0)
    flip = rng.random((nchunks, 1, ndims)) < 0.4                 # some columns never change sign: pure acceleration
    sign = np.where(flip, np.abs(sign), sign)
    delta = amp * sign * (1.0 + growth * (t % 16) / 16.0) + rng.integers(-2, 3, size=(nchunks, rows, ndims))
    x = np.cumsum(np.rint(delta).astype(np.int64), axis=1)
    return (x & ((1 << w) - 1)).astype(np.uint8 if w == 8 else np.uint16)

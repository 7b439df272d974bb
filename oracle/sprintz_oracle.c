/*
 * sprintz_oracle.c -- CPU restatement of the Sprintz codec hot path.
 * TEST INFRASTRUCTURE ONLY: parity oracle + portable CPU baseline.  See
 * sprintz_oracle.h for scope, citations and parity status (PINNED against the
 * compiled reference and tests/golden/).
 */
#include "sprintz_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- helpers */

static inline unsigned bitlen32(uint32_t v)
{
    unsigned n = 0;
    while (v) { n++; v >>= 1; }
    return n;
}

/* OR the low n (<=16) bits of v into a zeroed byte buffer, LSB-first. */
static inline void put_bits(uint8_t* base, uint64_t bitpos, uint32_t v, unsigned n)
{
    if (!n) return;
    v &= (1u << n) - 1u;
    uint64_t byte = bitpos >> 3;
    uint32_t x = v << (bitpos & 7);          /* <= 23 significant bits */
    while (x) { base[byte++] |= (uint8_t)x; x >>= 8; }
}

/* Read n (<=16) bits LSB-first; touches only the bytes that hold them. */
static inline uint32_t get_bits(const uint8_t* base, uint64_t bitpos, unsigned n)
{
    if (!n) return 0;
    uint64_t byte = bitpos >> 3;
    unsigned sh = (unsigned)(bitpos & 7);
    unsigned nbytes = (sh + n + 7) >> 3;     /* 1..3 */
    uint32_t x = 0;
    for (unsigned k = 0; k < nbytes; k++) x |= (uint32_t)base[byte + k] << (8 * k);
    return (x >> sh) & ((1u << n) - 1u);
}

/* format.h:36-45 */
static inline void put_container_header(uint8_t* out, uint32_t ngroups, uint16_t remaining, uint16_t ndims)
{
    memcpy(out, &ngroups, 4);
    memcpy(out + 4, &remaining, 2);
    memcpy(out + 6, &ndims, 2);
}

/* run length in blocks, 1 or 2 bytes (sprintz_xff_rle.cpp:377-384) */
static inline uint8_t* put_run_length(uint8_t* o, unsigned run)
{
    *o = (uint8_t)(run & 0x7f);
    if (run > 0x7f) { *o++ |= 0x80; *o++ = (uint8_t)(run >> 7); }
    else o++;
    return o;
}

/* ------------------------------------------------------------ instantiate */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

#define W 8
#define UINT_T uint8_t
#define INT_T int8_t
#define CTR_T int16_t
#define HB 3
#define SFX(x) CAT(x, _w8)
#include "sprintz_oracle_impl.inc"
#undef W
#undef UINT_T
#undef INT_T
#undef CTR_T
#undef HB
#undef SFX

#define W 16
#define UINT_T uint16_t
#define INT_T int16_t
#define CTR_T int32_t
#define HB 4
#define SFX(x) CAT(x, _w16)
#include "sprintz_oracle_impl.inc"
#undef W
#undef UINT_T
#undef INT_T
#undef CTR_T
#undef HB
#undef SFX

/* ------------------------------------------------------------- public API */

int64_t oracle_compress_ws(int codec, int elem_bytes, const void* src, uint32_t len, void* dest,
                           uint16_t ndims, int write_size, size_t* nbytes_out);

int64_t oracle_compress_ws(int codec, int elem_bytes, const void* src, uint32_t len, void* dest,
                           uint16_t ndims, int write_size, size_t* nbytes_out)
{
    if (ndims == 0) return -1;                              /* sprintz.cpp:36 */
    const int fire = codec != 0;
    if (elem_bytes == 1) {
        const int lowdim = ndims <= 4;                      /* sprintz.cpp:34-41 */
        return compress_w8((const uint8_t*)src, len, (uint8_t*)dest, ndims, fire, lowdim, write_size, nbytes_out);
    }
    const int lowdim = ndims <= 2;                          /* sprintz.cpp:43-50 */
    return compress_w16((const uint16_t*)src, len, (uint8_t*)dest, ndims, fire, lowdim, write_size, nbytes_out);
}

int64_t oracle_compress(int codec, int elem_bytes, const void* src, uint32_t len, void* dest,
                        uint16_t ndims, size_t* nbytes_out)
{
    return oracle_compress_ws(codec, elem_bytes, src, len, dest, ndims, 1, nbytes_out);
}

int64_t oracle_decompress_ex(int codec, int elem_bytes, const void* src, void* dest, int quirk,
                             size_t* consumed_bytes)
{
    const int fire = codec != 0;
    if (elem_bytes == 1) return decompress_w8((const uint8_t*)src, (uint8_t*)dest, fire, quirk, 0, consumed_bytes);
    return decompress_w16((const uint8_t*)src, (uint16_t*)dest, fire, quirk, 0, consumed_bytes);
}

/* ---- the reference's *_rowmajor_*_rle_* family: general row-major layout for EVERY ndims
 * (compress_rowmajor_delta_rle_8b sprintz_delta.h:49, ..._xff_rle_16b sprintz_xff.h:53); these are
 * what its query tests pair with query_rowmajor_* (test/test_query.cpp:59-120). */
int64_t oracle_compress_rowmajor(int codec, int elem_bytes, const void* src, uint32_t len, void* dest,
                                 uint16_t ndims, size_t* nbytes_out)
{
    if (ndims == 0) return -1;
    const int fire = codec != 0;
    if (elem_bytes == 1) return compress_w8((const uint8_t*)src, len, (uint8_t*)dest, ndims, fire, 0, 1, nbytes_out);
    return compress_w16((const uint16_t*)src, len, (uint8_t*)dest, ndims, fire, 0, 1, nbytes_out);
}

int64_t oracle_decompress_rowmajor_ex(int codec, int elem_bytes, const void* src, void* dest, int quirk,
                                      size_t* consumed_bytes)
{
    const int fire = codec != 0;
    if (elem_bytes == 1) return decompress_w8((const uint8_t*)src, (uint8_t*)dest, fire, quirk, 1, consumed_bytes);
    return decompress_w16((const uint8_t*)src, (uint16_t*)dest, fire, quirk, 1, consumed_bytes);
}

/* ---- non-RLE codecs (sprintz_delta.cpp:64-1391): raw = 1 bit-packing only (compress_rowmajor_{8b,16b}),
 * raw = 0 delta + bit-packing (compress_rowmajor_delta_{8b,16b}) */
int64_t oracle_compress_norle(int raw, int elem_bytes, const void* src, uint32_t len, void* dest, uint16_t ndims, size_t* nbytes_out)
{
    if (elem_bytes == 1) return compress_norle_w8((const uint8_t*)src, len, (uint8_t*)dest, ndims, raw, nbytes_out);
    return compress_norle_w16((const uint16_t*)src, len, (uint8_t*)dest, ndims, raw, nbytes_out);
}

int64_t oracle_decompress_norle(int raw, int elem_bytes, const void* src, void* dest, size_t* consumed_bytes)
{
    if (elem_bytes == 1) return decompress_norle_w8((const uint8_t*)src, (uint8_t*)dest, raw, consumed_bytes);
    return decompress_norle_w16((const uint8_t*)src, (uint16_t*)dest, raw, consumed_bytes);
}

/* ---- query on compressed data (query.hpp:23-29).  The reference throws its reduction
 * away (sprintz_xff_rle_query.cpp:69-104) and its functors are unfinished (query.hpp:222,
 * :84-87), so the RESULT is defined here, by restating what the operation means: op over the
 * decompressed data, per column (element index mod ndims), tail included, unsigned, 64-bit.
 * op 1 = max, 2 = sum.  `dest` receives the decompressed data (the materialised output). */
int64_t oracle_query(int codec, int elem_bytes, const void* src, void* dest, int general, int op, uint64_t* result)
{
    uint16_t D;
    memcpy(&D, (const uint8_t*)src + 6, 2);
    const int64_t n = general ? oracle_decompress_rowmajor_ex(codec, elem_bytes, src, dest, 0, NULL)
                              : oracle_decompress_ex(codec, elem_bytes, src, dest, 0, NULL);
    if (n < 0 || D == 0) return n;
    for (unsigned d = 0; d < D; d++) result[d] = 0;
    for (int64_t i = 0; i < n; i++) {
        const uint64_t x = elem_bytes == 1 ? ((const uint8_t*)dest)[i] : ((const uint16_t*)dest)[i];
        uint64_t* r = &result[i % D];
        if (op == 1) { if (x > *r) *r = x; }
        else if (op == 2) *r += x;
    }
    return n;
}

int64_t oracle_decompress_q(int codec, int elem_bytes, const void* src, void* dest, int quirk)
{
    return oracle_decompress_ex(codec, elem_bytes, src, dest, quirk, NULL);
}

int64_t oracle_decompress(int codec, int elem_bytes, const void* src, void* dest)
{
    return oracle_decompress_q(codec, elem_bytes, src, dest, 0);
}

int64_t oracle_compress_delta_8b(const uint8_t* s, uint32_t n, int8_t* d, uint16_t nd, size_t* nb)   { return oracle_compress(0, 1, s, n, d, nd, nb); }
int64_t oracle_compress_xff_8b(const uint8_t* s, uint32_t n, int8_t* d, uint16_t nd, size_t* nb)     { return oracle_compress(1, 1, s, n, d, nd, nb); }
int64_t oracle_compress_delta_16b(const uint16_t* s, uint32_t n, int16_t* d, uint16_t nd, size_t* nb){ return oracle_compress(0, 2, s, n, d, nd, nb); }
int64_t oracle_compress_xff_16b(const uint16_t* s, uint32_t n, int16_t* d, uint16_t nd, size_t* nb)  { return oracle_compress(1, 2, s, n, d, nd, nb); }

int64_t oracle_decompress_delta_8b(const int8_t* s, uint8_t* d)    { return oracle_decompress(0, 1, s, d); }
int64_t oracle_decompress_xff_8b(const int8_t* s, uint8_t* d)      { return oracle_decompress(1, 1, s, d); }
int64_t oracle_decompress_delta_16b(const int16_t* s, uint16_t* d) { return oracle_decompress(0, 2, s, d); }
int64_t oracle_decompress_xff_16b(const int16_t* s, uint16_t* d)   { return oracle_decompress(1, 2, s, d); }

/* Worst case: 8-byte container header; every group = header + 2 blocks of 8
 * rows x ceil(D*W/8) bytes (general) -- never more than raw + header per group;
 * plus the raw tail.  Slack of 16 bytes for callers that over-read/write. */
size_t oracle_compress_bound(int elem_bytes, uint32_t len, uint16_t ndims)
{
    if (ndims == 0) return 8;
    const size_t D = ndims, esz = (size_t)elem_bytes;
    const size_t hb = elem_bytes == 1 ? 3 : 4;
    const size_t group_elems = 16 * D;
    const size_t max_groups = len / group_elems + 1;
    const size_t hdr_bytes = (2 * D * hb + 7) / 8;
    return 8 + max_groups * (hdr_bytes + 2) + (size_t)len * esz + 16;
}

uint64_t oracle_compress_chunks(int codec, int elem_bytes, const void* src, uint64_t total_len,
                                uint32_t chunk_len, uint16_t ndims,
                                uint8_t* dest, size_t dest_stride, uint32_t* sizes)
{
    const uint8_t* s = (const uint8_t*)src;
    uint64_t total = 0, c = 0;
    for (uint64_t off = 0; off < total_len; off += chunk_len, c++) {
        uint32_t n = (uint32_t)((total_len - off < chunk_len) ? (total_len - off) : chunk_len);
        size_t nb = 0;
        oracle_compress(codec, elem_bytes, s + off * (uint64_t)elem_bytes, n, dest + c * dest_stride, ndims, &nb);
        sizes[c] = (uint32_t)nb;
        total += nb;
    }
    return total;
}

uint64_t oracle_decompress_chunks(int codec, int elem_bytes, const uint8_t* comp,
                                  const uint64_t* offsets, uint64_t nchunks,
                                  uint32_t chunk_len, void* out)
{
    uint8_t* o = (uint8_t*)out;
    uint64_t total = 0;
    for (uint64_t c = 0; c < nchunks; c++) {
        int64_t n = oracle_decompress(codec, elem_bytes, comp + offsets[c],
                                      o + c * (uint64_t)chunk_len * (uint64_t)elem_bytes);
        if (n > 0) total += (uint64_t)n;
    }
    return total;
}

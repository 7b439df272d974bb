/*
 * ref_shim.cpp -- C-ABI veneer over the REAL reference (dblalock/sprintz).
 * TEST INFRASTRUCTURE ONLY.
 *
 * This file contains no codec logic.  It is compiled together with the
 * reference's own sources *where they lie* under /root/reference/cpp/Compress
 * (see oracle/Makefile, target _ref) into oracle/_ref/libsprintz_ref.so, which
 * is used (a) to pin the CPU restatement and mint tests/golden/, and (b) as
 * the "reference" CPU baseline in bench.py.  Nothing under oracle/_ref/ is
 * committed; no reference source is copied into this repository.
 */
#include <stdint.h>
#include <stddef.h>
#include "sprintz.h"   /* -I/root/reference/cpp/Compress : sprintz.h:16-32 */
#include "sprintz_delta.h"   /* compress_rowmajor_delta_rle_*: :49-51,68-70; query_rowmajor_delta_rle_*: :95-98 */
#include "sprintz_xff.h"     /* compress_rowmajor_xff_rle_*: :45-55; query_rowmajor_xff_rle_*: :90-93 */
#include "delta.h"           /* encode/decode_{delta,doubledelta}_rowmajor_{8b,16b}: :17-68 */
#include "predict.h"         /* encode/decode_xff_rowmajor_{8b,16b}: :15-30 */
#include "online.hpp"        /* dynamic_delta_pack_u16 / zigzag_pack_u16 / sprintzpack_pack_u16 ...: :395-445 */

extern "C" {

int64_t ref_compress(int codec, int elem_bytes, const void* src, uint32_t len, void* dest,
                     uint16_t ndims, int write_size)
{
    const bool ws = write_size != 0;
    if (elem_bytes == 1) {
        return codec ? sprintz_compress_xff_8b((const uint8_t*)src, len, (int8_t*)dest, ndims, ws)
                     : sprintz_compress_delta_8b((const uint8_t*)src, len, (int8_t*)dest, ndims, ws);
    }
    return codec ? sprintz_compress_xff_16b((const uint16_t*)src, len, (int16_t*)dest, ndims, ws)
                 : sprintz_compress_delta_16b((const uint16_t*)src, len, (int16_t*)dest, ndims, ws);
}

int64_t ref_decompress(int codec, int elem_bytes, const void* src, void* dest)
{
    if (elem_bytes == 1) {
        return codec ? sprintz_decompress_xff_8b((const int8_t*)src, (uint8_t*)dest)
                     : sprintz_decompress_delta_8b((const int8_t*)src, (uint8_t*)dest);
    }
    return codec ? sprintz_decompress_xff_16b((const int16_t*)src, (uint16_t*)dest)
                 : sprintz_decompress_delta_16b((const int16_t*)src, (uint16_t*)dest);
}

/* chunk loops so that the CPU baseline is timed without per-call FFI overhead */
uint64_t ref_compress_chunks(int codec, int elem_bytes, const void* src, uint64_t total_len,
                             uint32_t chunk_len, uint16_t ndims,
                             uint8_t* dest, size_t dest_stride, int64_t* ret_elems)
{
    const uint8_t* s = (const uint8_t*)src;
    uint64_t c = 0, total = 0;
    for (uint64_t off = 0; off < total_len; off += chunk_len, c++) {
        uint32_t n = (uint32_t)((total_len - off < chunk_len) ? (total_len - off) : chunk_len);
        int64_t r = ref_compress(codec, elem_bytes, s + off * (uint64_t)elem_bytes, n,
                                 dest + c * dest_stride, ndims, 1);
        if (ret_elems) ret_elems[c] = r;
        if (r > 0) total += (uint64_t)r;
    }
    return total;
}

uint64_t ref_decompress_chunks(int codec, int elem_bytes, const uint8_t* comp,
                               const uint64_t* offsets, uint64_t nchunks,
                               uint32_t chunk_len, void* out)
{
    uint8_t* o = (uint8_t*)out;
    uint64_t total = 0;
    for (uint64_t c = 0; c < nchunks; c++) {
        int64_t n = ref_decompress(codec, elem_bytes, comp + offsets[c],
                                   o + c * (uint64_t)chunk_len * (uint64_t)elem_bytes);
        if (n > 0) total += (uint64_t)n;
    }
    return total;
}

/* the general row-major layout for every ndims (what the reference's query tests feed
 * query_rowmajor_*: test/test_query.cpp:59-120) */
int64_t ref_compress_rowmajor(int codec, int elem_bytes, const void* src, uint32_t len, void* dest, uint16_t ndims)
{
    if (elem_bytes == 1) {
        return codec ? compress_rowmajor_xff_rle_8b((const uint8_t*)src, len, (int8_t*)dest, ndims, true)
                     : compress_rowmajor_delta_rle_8b((const uint8_t*)src, len, (int8_t*)dest, ndims, true);
    }
    return codec ? compress_rowmajor_xff_rle_16b((const uint16_t*)src, len, (int16_t*)dest, ndims, true)
                 : compress_rowmajor_delta_rle_16b((const uint16_t*)src, len, (int16_t*)dest, ndims, true);
}

int64_t ref_decompress_rowmajor(int codec, int elem_bytes, const void* src, void* dest)
{
    if (elem_bytes == 1) {
        return codec ? decompress_rowmajor_xff_rle_8b((const int8_t*)src, (uint8_t*)dest)
                     : decompress_rowmajor_delta_rle_8b((const int8_t*)src, (uint8_t*)dest);
    }
    return codec ? decompress_rowmajor_xff_rle_16b((const int16_t*)src, (uint16_t*)dest)
                 : decompress_rowmajor_delta_rle_16b((const int16_t*)src, (uint16_t*)dest);
}

/* query with materialize (the only observable output of the reference's query path) */
int64_t ref_query(int codec, int elem_bytes, const void* src, void* dest, int op, int materialize)
{
    QueryParams qp;
    qp.op = (QueryTypes::Operation)op;
    qp.materialize = materialize != 0;
    if (elem_bytes == 1) {
        return codec ? query_rowmajor_xff_rle_8b((const int8_t*)src, (uint8_t*)dest, qp)
                     : query_rowmajor_delta_rle_8b((const int8_t*)src, (uint8_t*)dest, qp);
    }
    return codec ? query_rowmajor_xff_rle_16b((const int16_t*)src, (uint16_t*)dest, qp)
                 : query_rowmajor_delta_rle_16b((const int16_t*)src, (uint16_t*)dest, qp);
}

/* stand-alone transforms (delta.h:17-68, predict.h:15-30); kind 0 = delta, 1 = double delta, 2 = xff */
uint32_t ref_transform_encode(int kind, int elem_bytes, const void* src, uint32_t len, void* dest, uint16_t ndims, int write_size)
{
    const bool ws = write_size != 0;
    if (kind == 2) {
        return elem_bytes == 1 ? encode_xff_rowmajor_8b((const uint8_t*)src, len, (int8_t*)dest, ndims, ws)
                               : encode_xff_rowmajor_16b((const uint16_t*)src, len, (int16_t*)dest, ndims, ws);
    }
    if (elem_bytes == 1) {
        return kind ? encode_doubledelta_rowmajor_8b((const uint8_t*)src, len, (int8_t*)dest, ndims, ws)
                    : encode_delta_rowmajor_8b((const uint8_t*)src, len, (int8_t*)dest, ndims, ws);
    }
    return kind ? encode_doubledelta_rowmajor_16b((const uint16_t*)src, len, (int16_t*)dest, ndims, ws)
                : encode_delta_rowmajor_16b((const uint16_t*)src, len, (int16_t*)dest, ndims, ws);
}

/* src carries the 6-byte header (format.h:65-86) */
uint32_t ref_transform_decode(int kind, int elem_bytes, const void* src, void* dest)
{
    if (kind == 2) {
        return elem_bytes == 1 ? decode_xff_rowmajor_8b((const int8_t*)src, (uint8_t*)dest)
                               : decode_xff_rowmajor_16b((const int16_t*)src, (uint16_t*)dest);
    }
    if (elem_bytes == 1) {
        return kind ? decode_doubledelta_rowmajor_8b((const int8_t*)src, (uint8_t*)dest)
                    : decode_delta_rowmajor_8b((const int8_t*)src, (uint8_t*)dest);
    }
    return kind ? decode_doubledelta_rowmajor_16b((const int16_t*)src, (uint16_t*)dest)
                : decode_delta_rowmajor_16b((const int16_t*)src, (uint16_t*)dest);
}

/* non-RLE codecs (sprintz_delta.h): raw = 1 compress_rowmajor_{8b,16b}, raw = 0 compress_rowmajor_delta_{8b,16b} */
int64_t ref_compress_norle(int raw, int elem_bytes, const void* src, uint32_t len, void* dest, uint16_t ndims)
{
    if (raw == 2) return compress8b_rowmajor_xff((const uint8_t*)src, len, (int8_t*)dest, ndims, true);   /* sprintz_xff.h:28 */
    if (elem_bytes == 1) {
        return raw ? compress_rowmajor_8b((const uint8_t*)src, len, (int8_t*)dest, ndims, true)
                   : compress_rowmajor_delta_8b((const uint8_t*)src, len, (int8_t*)dest, ndims, true);
    }
    return raw ? compress_rowmajor_16b((const uint16_t*)src, len, (int16_t*)dest, ndims, true)
               : compress_rowmajor_delta_16b((const uint16_t*)src, len, (int16_t*)dest, ndims, true);
}

int64_t ref_decompress_norle(int raw, int elem_bytes, const void* src, void* dest)
{
    if (raw == 2) return decompress8b_rowmajor_xff((const int8_t*)src, (uint8_t*)dest);
    if (elem_bytes == 1) {
        return raw ? decompress_rowmajor_8b((const int8_t*)src, (uint8_t*)dest)
                   : decompress_rowmajor_delta_8b((const int8_t*)src, (uint8_t*)dest);
    }
    return raw ? decompress_rowmajor_16b((const int16_t*)src, (uint16_t*)dest)
               : decompress_rowmajor_delta_16b((const int16_t*)src, (uint16_t*)dest);
}


/* online.hpp's u16 coders; kind: 0 dynamic delta, 1 dynamic delta (alt loss), 2 zigzag, 3 sprintzpack, 4 sprintzpack + zigzag */
int64_t ref_online_pack(int kind, const uint16_t* src, uint32_t len, int16_t* dest)
{
    switch (kind) {
    case 0: return dynamic_delta_pack_u16(src, len, dest);
    case 1: return dynamic_delta_pack_u16_altloss(src, len, dest);
    case 2: return zigzag_pack_u16(src, len, dest);
    case 3: return sprintzpack_pack_u16(src, len, dest);
    case 4: return sprintzpack_pack_u16_zigzag(src, len, dest);
    }
    return -1;
}
int64_t ref_online_unpack(int kind, const int16_t* src, uint16_t* dest)
{
    switch (kind) {
    case 0: case 1: return dynamic_delta_unpack_u16(src, dest);
    case 2: return zigzag_unpack_u16(src, dest);
    case 3: return sprintzpack_unpack_u16(src, dest);
    case 4: return sprintzpack_unpack_u16_zigzag(src, dest);
    }
    return -1;
}
}

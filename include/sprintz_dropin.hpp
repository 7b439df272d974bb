// sprintz_dropin.hpp -- the reference's public header, re-declared on top of
// libsprintz_mi355x.so.  A caller written against dblalock/sprintz
// cpp/Compress/sprintz.h (sprintz.h:16-32; e.g. the author's lzbench fork,
// reference README.md:29) includes this file instead and links
// -lsprintz_mi355x: same names, same C++ signatures (default argument
// included), same units (ELEMENTS) and return values.
//
// Host pointers in and out; every call is one chunk on the GPU (see
// include/sprintz_mi355x.h for the batched device API that is actually fast).
#ifndef SPRINTZ_DROPIN_HPP
#define SPRINTZ_DROPIN_HPP

#include <stdint.h>

#include "sprintz_mi355x.h"

// ================================================================ 8b  (sprintz.h:16-23)
inline int64_t sprintz_compress_delta_8b(const uint8_t* src, uint32_t len, int8_t* dest, uint16_t ndims,
                                         bool write_size = true)
{
    return sprintz_mi355x_compress_delta_8b(src, len, dest, ndims, write_size ? 1 : 0);
}
inline int64_t sprintz_decompress_delta_8b(const int8_t* src, uint8_t* dest)
{
    return sprintz_mi355x_decompress_delta_8b(src, dest);
}
inline int64_t sprintz_compress_xff_8b(const uint8_t* src, uint32_t len, int8_t* dest, uint16_t ndims,
                                       bool write_size = true)
{
    return sprintz_mi355x_compress_xff_8b(src, len, dest, ndims, write_size ? 1 : 0);
}
inline int64_t sprintz_decompress_xff_8b(const int8_t* src, uint8_t* dest)
{
    return sprintz_mi355x_decompress_xff_8b(src, dest);
}

// ================================================================ 16b (sprintz.h:25-32)
inline int64_t sprintz_compress_delta_16b(const uint16_t* src, uint32_t len, int16_t* dest, uint16_t ndims,
                                          bool write_size = true)
{
    return sprintz_mi355x_compress_delta_16b(src, len, dest, ndims, write_size ? 1 : 0);
}
inline int64_t sprintz_decompress_delta_16b(const int16_t* src, uint16_t* dest)
{
    return sprintz_mi355x_decompress_delta_16b(src, dest);
}
inline int64_t sprintz_compress_xff_16b(const uint16_t* src, uint32_t len, int16_t* dest, uint16_t ndims,
                                        bool write_size = true)
{
    return sprintz_mi355x_compress_xff_16b(src, len, dest, ndims, write_size ? 1 : 0);
}
inline int64_t sprintz_decompress_xff_16b(const int16_t* src, uint16_t* dest)
{
    return sprintz_mi355x_decompress_xff_16b(src, dest);
}

// ================================================================ query on compressed data
// query.hpp:23-29 (QueryTypes, QueryParams); sprintz_delta.h:95-98, sprintz_xff.h:90-93.  Same names
// and signatures; the overloads with `result` (ndims uint64) return what the reference computes
// and discards (see include/sprintz_mi355x.h for the definition).  These functions take
// streams in the general row-major layout whatever ndims is, like the reference's.
namespace QueryTypes {
enum Operation { NOOP = 0, REDUCE_MAX, REDUCE_SUM };
}
typedef struct QueryParams {
    QueryTypes::Operation op;
    bool materialize;
} QueryParams;

inline int64_t query_rowmajor_delta_rle_8b(const int8_t* src, uint8_t* dest, const QueryParams& qp, uint64_t* result = nullptr)
{
    return sprintz_mi355x_query_delta_8b(src, dest, (int)qp.op, qp.materialize ? 1 : 0, SPRINTZ_QUERY_GENERAL_LAYOUT, result);
}
inline int64_t query_rowmajor_delta_rle_16b(const int16_t* src, uint16_t* dest, const QueryParams& qp, uint64_t* result = nullptr)
{
    return sprintz_mi355x_query_delta_16b(src, dest, (int)qp.op, qp.materialize ? 1 : 0, SPRINTZ_QUERY_GENERAL_LAYOUT, result);
}
inline int64_t query_rowmajor_xff_rle_8b(const int8_t* src, uint8_t* dest, const QueryParams& qp, uint64_t* result = nullptr)
{
    return sprintz_mi355x_query_xff_8b(src, dest, (int)qp.op, qp.materialize ? 1 : 0, SPRINTZ_QUERY_GENERAL_LAYOUT, result);
}
inline int64_t query_rowmajor_xff_rle_16b(const int16_t* src, uint16_t* dest, const QueryParams& qp, uint64_t* result = nullptr)
{
    return sprintz_mi355x_query_xff_16b(src, dest, (int)qp.op, qp.materialize ? 1 : 0, SPRINTZ_QUERY_GENERAL_LAYOUT, result);
}

// ================================================================ non-RLE codecs (sprintz_delta.h:26-76)
inline int64_t compress_rowmajor_8b(const uint8_t* src, uint32_t len, int8_t* dest, uint16_t ndims, bool = true)
{
    return sprintz_mi355x_compress_norle(SPRINTZ_CODEC_BITPACK_NORLE, 1, src, len, dest, ndims);
}
inline int64_t compress_rowmajor_16b(const uint16_t* src, uint32_t len, int16_t* dest, uint16_t ndims, bool = true)
{
    return sprintz_mi355x_compress_norle(SPRINTZ_CODEC_BITPACK_NORLE, 2, src, len, dest, ndims);
}
inline int64_t compress_rowmajor_delta_8b(const uint8_t* src, uint32_t len, int8_t* dest, uint16_t ndims, bool = true)
{
    return sprintz_mi355x_compress_norle(SPRINTZ_CODEC_DELTA_NORLE, 1, src, len, dest, ndims);
}
inline int64_t compress_rowmajor_delta_16b(const uint16_t* src, uint32_t len, int16_t* dest, uint16_t ndims, bool = true)
{
    return sprintz_mi355x_compress_norle(SPRINTZ_CODEC_DELTA_NORLE, 2, src, len, dest, ndims);
}
inline int64_t decompress_rowmajor_8b(const int8_t* src, uint8_t* dest) { return sprintz_mi355x_decompress_norle(SPRINTZ_CODEC_BITPACK_NORLE, 1, src, dest); }
inline int64_t decompress_rowmajor_16b(const int16_t* src, uint16_t* dest) { return sprintz_mi355x_decompress_norle(SPRINTZ_CODEC_BITPACK_NORLE, 2, src, dest); }
inline int64_t decompress_rowmajor_delta_8b(const int8_t* src, uint8_t* dest) { return sprintz_mi355x_decompress_norle(SPRINTZ_CODEC_DELTA_NORLE, 1, src, dest); }
inline int64_t decompress_rowmajor_delta_16b(const int16_t* src, uint16_t* dest) { return sprintz_mi355x_decompress_norle(SPRINTZ_CODEC_DELTA_NORLE, 2, src, dest); }

inline int64_t compress8b_rowmajor_xff(const uint8_t* src, uint64_t len, int8_t* dest, uint16_t ndims, bool = true)   // sprintz_xff.h:28
{
    return len >> 32 ? -1 : sprintz_mi355x_compress_norle(SPRINTZ_CODEC_XFF_NORLE, 1, src, (uint32_t)len, dest, ndims);
}
inline int64_t decompress8b_rowmajor_xff(const int8_t* src, uint8_t* dest) { return sprintz_mi355x_decompress_norle(SPRINTZ_CODEC_XFF_NORLE, 1, src, dest); }

// ================================================================ stand-alone transforms (delta.h:17-68, predict.h:15-30)
#define SPRINTZ_DROPIN_TRANSFORM(NAME, KIND, BITS, ESZ)                                                                       \
    inline uint32_t encode_##NAME##_rowmajor_##BITS##b(const uint##BITS##_t* src, uint32_t len, int##BITS##_t* dest, uint16_t ndims,  \
                                                       bool write_size = true)                                                \
    {                                                                                                                         \
        return (uint32_t)sprintz_mi355x_transform_encode(KIND, ESZ, src, len, dest, ndims, write_size ? 1 : 0);               \
    }                                                                                                                         \
    inline uint32_t decode_##NAME##_rowmajor_##BITS##b(const int##BITS##_t* src, uint32_t len, uint##BITS##_t* dest, uint16_t ndims)  \
    {                                                                                                                         \
        return ndims == 0 ? 0u : (uint32_t)sprintz_mi355x_transform_decode(KIND, ESZ, src, dest, len, ndims);                 \
    }                                                                                                                         \
    inline uint32_t decode_##NAME##_rowmajor_##BITS##b(const int##BITS##_t* src, uint##BITS##_t* dest)                        \
    {                                                                                                                         \
        return (uint32_t)sprintz_mi355x_transform_decode(KIND, ESZ, src, dest, 0, 0);                                         \
    }                                                                                                                         \
    inline uint32_t decode_##NAME##_rowmajor_inplace_##BITS##b(uint##BITS##_t* buff, uint32_t len, uint16_t ndims)            \
    {   /* the device copy is the temporary the reference mallocs (delta.cpp:351-373) */                                      \
        return ndims == 0 ? 0u : (uint32_t)sprintz_mi355x_transform_decode(KIND, ESZ, buff, buff, len, ndims);                \
    }
SPRINTZ_DROPIN_TRANSFORM(delta, SPRINTZ_TRANSFORM_DELTA, 8, 1)
SPRINTZ_DROPIN_TRANSFORM(delta, SPRINTZ_TRANSFORM_DELTA, 16, 2)
SPRINTZ_DROPIN_TRANSFORM(doubledelta, SPRINTZ_TRANSFORM_DOUBLEDELTA, 8, 1)
SPRINTZ_DROPIN_TRANSFORM(doubledelta, SPRINTZ_TRANSFORM_DOUBLEDELTA, 16, 2)
SPRINTZ_DROPIN_TRANSFORM(xff, SPRINTZ_TRANSFORM_XFF, 8, 1)
SPRINTZ_DROPIN_TRANSFORM(xff, SPRINTZ_TRANSFORM_XFF, 16, 2)
#undef SPRINTZ_DROPIN_TRANSFORM

#endif  // SPRINTZ_DROPIN_HPP

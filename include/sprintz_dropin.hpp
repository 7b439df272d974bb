// sprintz_dropin.hpp -- the reference's public prototypes, DECLARED here and DEFINED (non-inline,
// C++ linkage, the reference's exact signatures and therefore its exact mangled names) in
// libsprintz_mi355x.so (sprintz_amd/csrc/dropin.cpp).  An object file already compiled against
// dblalock/sprintz cpp/Compress/sprintz.h (sprintz.h:16-32; e.g. the author's lzbench fork,
// reference README.md:29) links against -lsprintz_mi355x unchanged; a caller being recompiled
// may include this file instead of the reference's headers: same names, same default arguments,
// same units (ELEMENTS) and return values.
//
// Host pointers in and out; every call is one chunk on the GPU (see include/sprintz_mi355x.h for
// the batched device API that is actually fast).  Error convention of the reference kept: the
// int64_t functions return -1 for ndims == 0 (sprintz.cpp:36) and otherwise a negative
// SPRINTZ_E_* code on failure (the reference has no other error path); the uint32_t transform
// functions return 0 on failure (sprintz_mi355x_last_error() says why).
#ifndef SPRINTZ_DROPIN_HPP
#define SPRINTZ_DROPIN_HPP

#include <stddef.h>
#include <stdint.h>

#include "sprintz_mi355x.h"

// ================================================================ 8b  (sprintz.h:16-23)
int64_t sprintz_compress_delta_8b(const uint8_t* src, uint32_t len, int8_t* dest, uint16_t ndims, bool write_size = true);
int64_t sprintz_decompress_delta_8b(const int8_t* src, uint8_t* dest);
int64_t sprintz_compress_xff_8b(const uint8_t* src, uint32_t len, int8_t* dest, uint16_t ndims, bool write_size = true);
int64_t sprintz_decompress_xff_8b(const int8_t* src, uint8_t* dest);

// ================================================================ 16b (sprintz.h:25-32)
int64_t sprintz_compress_delta_16b(const uint16_t* src, uint32_t len, int16_t* dest, uint16_t ndims, bool write_size = true);
int64_t sprintz_decompress_delta_16b(const int16_t* src, uint16_t* dest);
int64_t sprintz_compress_xff_16b(const uint16_t* src, uint32_t len, int16_t* dest, uint16_t ndims, bool write_size = true);
int64_t sprintz_decompress_xff_16b(const int16_t* src, uint16_t* dest);

// ================================================================ the layer below sprintz.h
// compress_rowmajor_{delta,xff}_rle[_lowdim]_{8b,16b} and their decoders (sprintz_delta.h:49-91,
// sprintz_xff.h:43-85): the same four codecs with the payload layout chosen by NAME instead of by
// ndims -- the plain names use the general row-major layout for every ndims, the _lowdim names the
// column-major low-dim layout (ndims <= 4 at 8 bits, <= 2 at 16 bits; else -1 as
// sprintz_delta_lowdim.cpp:64-70).  The 5-argument forms decode streams written with write_size = false.
#define SPRINTZ_DROPIN_RLE(NAME, BITS)                                                                                          \
    int64_t compress_rowmajor_##NAME##_##BITS##b(const uint##BITS##_t* src, uint32_t len, int##BITS##_t* dest, uint16_t ndims, \
                                                 bool write_size = true);                                                     \
    int64_t decompress_rowmajor_##NAME##_##BITS##b(const int##BITS##_t* src, uint##BITS##_t* dest);
SPRINTZ_DROPIN_RLE(delta_rle, 8)
SPRINTZ_DROPIN_RLE(delta_rle, 16)
SPRINTZ_DROPIN_RLE(xff_rle, 8)
SPRINTZ_DROPIN_RLE(xff_rle, 16)
SPRINTZ_DROPIN_RLE(delta_rle_lowdim, 8)
SPRINTZ_DROPIN_RLE(delta_rle_lowdim, 16)
SPRINTZ_DROPIN_RLE(xff_rle_lowdim, 8)
SPRINTZ_DROPIN_RLE(xff_rle_lowdim, 16)
#undef SPRINTZ_DROPIN_RLE
// (the reference's 5-argument decoders are force-inlined header code, sprintz_delta.h:52-54: no symbol to
//  replace; the equivalent entry point is sprintz_mi355x_decompress_noheader)

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

int64_t query_rowmajor_delta_rle_8b(const int8_t* src, uint8_t* dest, const QueryParams& qp);
int64_t query_rowmajor_delta_rle_16b(const int16_t* src, uint16_t* dest, const QueryParams& qp);
int64_t query_rowmajor_xff_rle_8b(const int8_t* src, uint8_t* dest, const QueryParams& qp);
int64_t query_rowmajor_xff_rle_16b(const int16_t* src, uint16_t* dest, const QueryParams& qp);
int64_t query_rowmajor_delta_rle_8b(const int8_t* src, uint8_t* dest, const QueryParams& qp, uint64_t* result);
int64_t query_rowmajor_delta_rle_16b(const int16_t* src, uint16_t* dest, const QueryParams& qp, uint64_t* result);
int64_t query_rowmajor_xff_rle_8b(const int8_t* src, uint8_t* dest, const QueryParams& qp, uint64_t* result);
int64_t query_rowmajor_xff_rle_16b(const int16_t* src, uint16_t* dest, const QueryParams& qp, uint64_t* result);

// ================================================================ non-RLE codecs (sprintz_delta.h:26-44, sprintz_xff.h:28-31)
int64_t compress_rowmajor_8b(const uint8_t* src, uint32_t len, int8_t* dest, uint16_t ndims, bool write_size = true);
int64_t compress_rowmajor_16b(const uint16_t* src, uint32_t len, int16_t* dest, uint16_t ndims, bool write_size = true);
int64_t compress_rowmajor_delta_8b(const uint8_t* src, uint32_t len, int8_t* dest, uint16_t ndims, bool write_size = true);
int64_t compress_rowmajor_delta_16b(const uint16_t* src, uint32_t len, int16_t* dest, uint16_t ndims, bool write_size = true);
int64_t decompress_rowmajor_8b(const int8_t* src, uint8_t* dest);
int64_t decompress_rowmajor_16b(const int16_t* src, uint16_t* dest);
int64_t decompress_rowmajor_delta_8b(const int8_t* src, uint8_t* dest);
int64_t decompress_rowmajor_delta_16b(const int16_t* src, uint16_t* dest);
int64_t compress8b_rowmajor_xff(const uint8_t* src, uint64_t len, int8_t* dest, uint16_t ndims, bool write_size = true);
int64_t decompress8b_rowmajor_xff(const int8_t* src, uint8_t* dest);

// ================================================================ stand-alone transforms (delta.h:17-68, predict.h:15-30)
#define SPRINTZ_DROPIN_TRANSFORM(NAME, BITS)                                                                                   \
    uint32_t encode_##NAME##_rowmajor_##BITS##b(const uint##BITS##_t* src, uint32_t len, int##BITS##_t* dest, uint16_t ndims, \
                                                bool write_size = true);                                                      \
    uint32_t decode_##NAME##_rowmajor_##BITS##b(const int##BITS##_t* src, uint32_t len, uint##BITS##_t* dest, uint16_t ndims); \
    uint32_t decode_##NAME##_rowmajor_##BITS##b(const int##BITS##_t* src, uint##BITS##_t* dest);                              \
    uint32_t decode_##NAME##_rowmajor_inplace_##BITS##b(uint##BITS##_t* buff, uint32_t len, uint16_t ndims);
SPRINTZ_DROPIN_TRANSFORM(delta, 8)
SPRINTZ_DROPIN_TRANSFORM(delta, 16)
SPRINTZ_DROPIN_TRANSFORM(doubledelta, 8)
SPRINTZ_DROPIN_TRANSFORM(doubledelta, 16)
SPRINTZ_DROPIN_TRANSFORM(xff, 8)
SPRINTZ_DROPIN_TRANSFORM(xff, 16)
#undef SPRINTZ_DROPIN_TRANSFORM

// ================================================================ online.hpp:395-445 (1-D uint16 streams)
typedef uint32_t len_t;                                    // online.hpp:15
len_t dynamic_delta_pack_u16(const uint16_t* data_in, size_t length, int16_t* data_out);
len_t dynamic_delta_pack_u16_altloss(const uint16_t* data_in, size_t length, int16_t* data_out);
len_t dynamic_delta_unpack_u16(const int16_t* data_in, uint16_t* data_out);
len_t zigzag_pack_u16(const uint16_t* data_in, size_t length, int16_t* data_out);
len_t zigzag_unpack_u16(const int16_t* data_in, uint16_t* data_out);
len_t sprintzpack_pack_u16(const uint16_t* data_in, size_t length, int16_t* data_out);
len_t sprintzpack_pack_u16_zigzag(const uint16_t* data_in, size_t length, int16_t* data_out);
len_t sprintzpack_unpack_u16(const int16_t* data_in, uint16_t* data_out);
len_t sprintzpack_unpack_u16_zigzag(const int16_t* data_in, uint16_t* data_out);

#endif  // SPRINTZ_DROPIN_HPP

// dropin.cpp -- the reference's public functions, defined with the reference's own C++ signatures
// (and therefore its own mangled names: _Z25sprintz_compress_delta_8bPKhjPatb ...) on top of the
// C-ABI.  No codec logic here: every function is a one-line hand-over to api.hip / transforms.hip.
// An object file compiled against dblalock/sprintz cpp/Compress/{sprintz,sprintz_delta,sprintz_xff,
// delta,predict}.h links against libsprintz_mi355x.so without recompilation.
#include "../../include/sprintz_dropin.hpp"

#define VIS __attribute__((visibility("default")))

// ---------------------------------------------------------------- sprintz.h:16-32
VIS int64_t sprintz_compress_delta_8b(const uint8_t* s, uint32_t n, int8_t* d, uint16_t nd, bool ws) { return sprintz_mi355x_compress_delta_8b(s, n, d, nd, ws ? 1 : 0); }
VIS int64_t sprintz_decompress_delta_8b(const int8_t* s, uint8_t* d) { return sprintz_mi355x_decompress_delta_8b(s, d); }
VIS int64_t sprintz_compress_xff_8b(const uint8_t* s, uint32_t n, int8_t* d, uint16_t nd, bool ws) { return sprintz_mi355x_compress_xff_8b(s, n, d, nd, ws ? 1 : 0); }
VIS int64_t sprintz_decompress_xff_8b(const int8_t* s, uint8_t* d) { return sprintz_mi355x_decompress_xff_8b(s, d); }
VIS int64_t sprintz_compress_delta_16b(const uint16_t* s, uint32_t n, int16_t* d, uint16_t nd, bool ws) { return sprintz_mi355x_compress_delta_16b(s, n, d, nd, ws ? 1 : 0); }
VIS int64_t sprintz_decompress_delta_16b(const int16_t* s, uint16_t* d) { return sprintz_mi355x_decompress_delta_16b(s, d); }
VIS int64_t sprintz_compress_xff_16b(const uint16_t* s, uint32_t n, int16_t* d, uint16_t nd, bool ws) { return sprintz_mi355x_compress_xff_16b(s, n, d, nd, ws ? 1 : 0); }
VIS int64_t sprintz_decompress_xff_16b(const int16_t* s, uint16_t* d) { return sprintz_mi355x_decompress_xff_16b(s, d); }

// ---------------------------------------------------------------- sprintz_delta.h:49-91, sprintz_xff.h:43-85
#define DROPIN_RLE(NAME, BITS, CODEC, LAYOUT)                                                                                   \
    VIS int64_t compress_rowmajor_##NAME##_##BITS##b(const uint##BITS##_t* s, uint32_t n, int##BITS##_t* d, uint16_t nd, bool ws) \
    {                                                                                                                           \
        return sprintz_mi355x_compress_layout(CODEC, BITS / 8, s, n, d, nd, ws ? 1 : 0, LAYOUT);                                \
    }                                                                                                                           \
    VIS int64_t decompress_rowmajor_##NAME##_##BITS##b(const int##BITS##_t* s, uint##BITS##_t* d)                               \
    {                                                                                                                           \
        return sprintz_mi355x_decompress_layout(CODEC, BITS / 8, s, d, LAYOUT);                                                 \
    }
DROPIN_RLE(delta_rle, 8, SPRINTZ_CODEC_DELTA, SPRINTZ_LAYOUT_GENERAL)
DROPIN_RLE(delta_rle, 16, SPRINTZ_CODEC_DELTA, SPRINTZ_LAYOUT_GENERAL)
DROPIN_RLE(xff_rle, 8, SPRINTZ_CODEC_XFF, SPRINTZ_LAYOUT_GENERAL)
DROPIN_RLE(xff_rle, 16, SPRINTZ_CODEC_XFF, SPRINTZ_LAYOUT_GENERAL)
DROPIN_RLE(delta_rle_lowdim, 8, SPRINTZ_CODEC_DELTA, SPRINTZ_LAYOUT_LOWDIM)
DROPIN_RLE(delta_rle_lowdim, 16, SPRINTZ_CODEC_DELTA, SPRINTZ_LAYOUT_LOWDIM)
DROPIN_RLE(xff_rle_lowdim, 8, SPRINTZ_CODEC_XFF, SPRINTZ_LAYOUT_LOWDIM)
DROPIN_RLE(xff_rle_lowdim, 16, SPRINTZ_CODEC_XFF, SPRINTZ_LAYOUT_LOWDIM)
#undef DROPIN_RLE

// ---------------------------------------------------------------- query.hpp:23-29, sprintz_delta.h:95-98, sprintz_xff.h:90-93
#define DROPIN_QUERY(NAME, BITS, FN)                                                                                            \
    VIS int64_t query_rowmajor_##NAME##_rle_##BITS##b(const int##BITS##_t* s, uint##BITS##_t* d, const QueryParams& qp, uint64_t* result) \
    {                                                                                                                           \
        return FN(s, d, (int)qp.op, qp.materialize ? 1 : 0, SPRINTZ_QUERY_GENERAL_LAYOUT, result);                              \
    }                                                                                                                           \
    VIS int64_t query_rowmajor_##NAME##_rle_##BITS##b(const int##BITS##_t* s, uint##BITS##_t* d, const QueryParams& qp)        \
    {                                                                                                                           \
        return FN(s, d, (int)qp.op, qp.materialize ? 1 : 0, SPRINTZ_QUERY_GENERAL_LAYOUT, nullptr);                             \
    }
DROPIN_QUERY(delta, 8, sprintz_mi355x_query_delta_8b)
DROPIN_QUERY(delta, 16, sprintz_mi355x_query_delta_16b)
DROPIN_QUERY(xff, 8, sprintz_mi355x_query_xff_8b)
DROPIN_QUERY(xff, 16, sprintz_mi355x_query_xff_16b)
#undef DROPIN_QUERY

// ---------------------------------------------------------------- sprintz_delta.h:26-44, sprintz_xff.h:28-31
// (these codecs always write their header: the reference's write_size argument only exists in the signature)
VIS int64_t compress_rowmajor_8b(const uint8_t* s, uint32_t n, int8_t* d, uint16_t nd, bool) { return sprintz_mi355x_compress_norle(SPRINTZ_CODEC_BITPACK_NORLE, 1, s, n, d, nd); }
VIS int64_t compress_rowmajor_16b(const uint16_t* s, uint32_t n, int16_t* d, uint16_t nd, bool) { return sprintz_mi355x_compress_norle(SPRINTZ_CODEC_BITPACK_NORLE, 2, s, n, d, nd); }
VIS int64_t compress_rowmajor_delta_8b(const uint8_t* s, uint32_t n, int8_t* d, uint16_t nd, bool) { return sprintz_mi355x_compress_norle(SPRINTZ_CODEC_DELTA_NORLE, 1, s, n, d, nd); }
VIS int64_t compress_rowmajor_delta_16b(const uint16_t* s, uint32_t n, int16_t* d, uint16_t nd, bool) { return sprintz_mi355x_compress_norle(SPRINTZ_CODEC_DELTA_NORLE, 2, s, n, d, nd); }
VIS int64_t decompress_rowmajor_8b(const int8_t* s, uint8_t* d) { return sprintz_mi355x_decompress_norle(SPRINTZ_CODEC_BITPACK_NORLE, 1, s, d); }
VIS int64_t decompress_rowmajor_16b(const int16_t* s, uint16_t* d) { return sprintz_mi355x_decompress_norle(SPRINTZ_CODEC_BITPACK_NORLE, 2, s, d); }
VIS int64_t decompress_rowmajor_delta_8b(const int8_t* s, uint8_t* d) { return sprintz_mi355x_decompress_norle(SPRINTZ_CODEC_DELTA_NORLE, 1, s, d); }
VIS int64_t decompress_rowmajor_delta_16b(const int16_t* s, uint16_t* d) { return sprintz_mi355x_decompress_norle(SPRINTZ_CODEC_DELTA_NORLE, 2, s, d); }
VIS int64_t compress8b_rowmajor_xff(const uint8_t* s, uint64_t n, int8_t* d, uint16_t nd, bool)
{
    return n >> 32 ? -1 : sprintz_mi355x_compress_norle(SPRINTZ_CODEC_XFF_NORLE, 1, s, (uint32_t)n, d, nd);
}
VIS int64_t decompress8b_rowmajor_xff(const int8_t* s, uint8_t* d) { return sprintz_mi355x_decompress_norle(SPRINTZ_CODEC_XFF_NORLE, 1, s, d); }

// ---------------------------------------------------------------- delta.h:17-68, predict.h:15-30
// (uint32_t returns: a failure reads as 0 elements, never as a negative code cast to a huge count)
static inline uint32_t count_or_zero(int64_t rc) { return rc < 0 ? 0u : (uint32_t)rc; }
#define DROPIN_TRANSFORM(NAME, KIND, BITS)                                                                                      \
    VIS uint32_t encode_##NAME##_rowmajor_##BITS##b(const uint##BITS##_t* s, uint32_t n, int##BITS##_t* d, uint16_t nd, bool ws)  \
    {                                                                                                                           \
        return count_or_zero(sprintz_mi355x_transform_encode(KIND, BITS / 8, s, n, d, nd, ws ? 1 : 0));                         \
    }                                                                                                                           \
    VIS uint32_t decode_##NAME##_rowmajor_##BITS##b(const int##BITS##_t* s, uint32_t n, uint##BITS##_t* d, uint16_t nd)         \
    {                                                                                                                           \
        return nd == 0 ? 0u : count_or_zero(sprintz_mi355x_transform_decode(KIND, BITS / 8, s, d, n, nd));                      \
    }                                                                                                                           \
    VIS uint32_t decode_##NAME##_rowmajor_##BITS##b(const int##BITS##_t* s, uint##BITS##_t* d)                                  \
    {                                                                                                                           \
        return count_or_zero(sprintz_mi355x_transform_decode(KIND, BITS / 8, s, d, 0, 0));                                      \
    }                                                                                                                           \
    VIS uint32_t decode_##NAME##_rowmajor_inplace_##BITS##b(uint##BITS##_t* buff, uint32_t n, uint16_t nd)                      \
    {   /* the device copy is the temporary the reference mallocs (delta.cpp:351-373) */                                        \
        return nd == 0 ? 0u : count_or_zero(sprintz_mi355x_transform_decode(KIND, BITS / 8, buff, buff, n, nd));                \
    }
DROPIN_TRANSFORM(delta, SPRINTZ_TRANSFORM_DELTA, 8)
DROPIN_TRANSFORM(delta, SPRINTZ_TRANSFORM_DELTA, 16)
DROPIN_TRANSFORM(doubledelta, SPRINTZ_TRANSFORM_DOUBLEDELTA, 8)
DROPIN_TRANSFORM(doubledelta, SPRINTZ_TRANSFORM_DOUBLEDELTA, 16)
DROPIN_TRANSFORM(xff, SPRINTZ_TRANSFORM_XFF, 8)
DROPIN_TRANSFORM(xff, SPRINTZ_TRANSFORM_XFF, 16)
#undef DROPIN_TRANSFORM

// ---------------------------------------------------------------- online.hpp:395-445
// (len_t returns: a failure reads as 0 elements)
VIS len_t dynamic_delta_pack_u16(const uint16_t* s, size_t n, int16_t* d) { return n >> 32 ? 0u : count_or_zero(sprintz_mi355x_online_pack(SPRINTZ_ONLINE_DYNDELTA, s, (uint32_t)n, d)); }
VIS len_t dynamic_delta_pack_u16_altloss(const uint16_t* s, size_t n, int16_t* d) { return n >> 32 ? 0u : count_or_zero(sprintz_mi355x_online_pack(SPRINTZ_ONLINE_DYNDELTA_ALT, s, (uint32_t)n, d)); }
VIS len_t dynamic_delta_unpack_u16(const int16_t* s, uint16_t* d) { return count_or_zero(sprintz_mi355x_online_unpack(SPRINTZ_ONLINE_DYNDELTA, s, d)); }
VIS len_t zigzag_pack_u16(const uint16_t* s, size_t n, int16_t* d) { return n >> 32 ? 0u : count_or_zero(sprintz_mi355x_online_pack(SPRINTZ_ONLINE_ZIGZAG, s, (uint32_t)n, d)); }
VIS len_t zigzag_unpack_u16(const int16_t* s, uint16_t* d) { return count_or_zero(sprintz_mi355x_online_unpack(SPRINTZ_ONLINE_ZIGZAG, s, d)); }
VIS len_t sprintzpack_pack_u16(const uint16_t* s, size_t n, int16_t* d) { return n >> 32 ? 0u : count_or_zero(sprintz_mi355x_online_pack(SPRINTZ_ONLINE_PACK, s, (uint32_t)n, d)); }
VIS len_t sprintzpack_pack_u16_zigzag(const uint16_t* s, size_t n, int16_t* d) { return n >> 32 ? 0u : count_or_zero(sprintz_mi355x_online_pack(SPRINTZ_ONLINE_PACK_ZIGZAG, s, (uint32_t)n, d)); }
VIS len_t sprintzpack_unpack_u16(const int16_t* s, uint16_t* d) { return count_or_zero(sprintz_mi355x_online_unpack(SPRINTZ_ONLINE_PACK, s, d)); }
VIS len_t sprintzpack_unpack_u16_zigzag(const int16_t* s, uint16_t* d) { return count_or_zero(sprintz_mi355x_online_unpack(SPRINTZ_ONLINE_PACK_ZIGZAG, s, d)); }

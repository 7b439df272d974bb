/*
 * transforms_oracle.c -- CPU restatement of the reference's stand-alone transforms
 * (SURVEY.md 8f-2).  TEST INFRASTRUCTURE ONLY.  Parity: PINNED against the compiled
 * reference (oracle/_ref, ref_transform_*) by tests/test_transforms_cpu.py and the fixtures
 * in tests/golden/golden_transforms_v1.npz.
 *
 *   kind 0  delta         encode_delta_rowmajor        cpp/Compress/delta.cpp:35-121
 *                         decode_delta_rowmajor        delta.cpp:133-232,309-397
 *   kind 1  double delta  encode_doubledelta_rowmajor  delta.cpp:405-465,532-610
 *                         decode_doubledelta_rowmajor  delta.cpp:467-529,623-693
 *
 * Container: 6-byte header {u32 len; u16 ndims} (format.h:65-86) = 6 elements at 8 bits,
 * 3 at 16, then `len` transformed elements.  Arithmetic wraps at the element width.
 *
 * What the vectorised reference code computes, stripped of its striping: per column c
 * (element index mod ndims), with the state starting at zero,
 *   delta:         y[r] = x[r] - x[r-1]
 *   double delta:  d[r] = x[r] - x[r-1];  y[r] = d[r] - d[r-1]
 * (the first row is "x - 0", i.e. copied; delta.cpp:107-111, :432-440).
 */
#include <stdint.h>
#include <string.h>

#define DEFINE(SFX, U)                                                                         \
    static void enc_##SFX(int kind, const U* x, uint32_t len, U* y, uint32_t D)               \
    {                                                                                          \
        for (uint32_t i = 0; i < len; i++) {                                                   \
            const U p1 = i >= D ? x[i - D] : 0, p2 = i >= 2 * D ? x[i - 2 * D] : 0;            \
            y[i] = kind ? (U)(x[i] - 2 * p1 + p2) : (U)(x[i] - p1);                            \
        }                                                                                      \
    }                                                                                          \
    static void dec_##SFX(int kind, const U* y, uint32_t len, U* x, uint32_t D)               \
    {                                                                                          \
        for (uint32_t i = 0; i < len; i++) {                                                   \
            const U p1 = i >= D ? x[i - D] : 0, p2 = i >= 2 * D ? x[i - 2 * D] : 0;            \
            x[i] = kind ? (U)(y[i] + 2 * p1 - p2) : (U)(y[i] + p1);                            \
        }                                                                                      \
    }
DEFINE(8, uint8_t)
DEFINE(16, uint16_t)

/* returns len + header length in elements, like the reference (delta.cpp:120, :609) */
uint32_t oracle_transform_encode(int kind, int elem_bytes, const void* src, uint32_t len, void* dest, uint16_t ndims, int write_size)
{
    uint8_t* d = (uint8_t*)dest;
    uint32_t hdr = 0;
    if (write_size) {
        memcpy(d, &len, 4);
        memcpy(d + 4, &ndims, 2);
        hdr = elem_bytes == 1 ? 6 : 3;
        d += 6;
    }
    if (ndims == 0) return len + hdr;
    if (elem_bytes == 1) enc_8(kind, (const uint8_t*)src, len, d, ndims);
    else enc_16(kind, (const uint16_t*)src, len, (uint16_t*)d, ndims);
    return len + hdr;
}

/* src carries the header; returns len */
uint32_t oracle_transform_decode(int kind, int elem_bytes, const void* src, void* dest)
{
    const uint8_t* s = (const uint8_t*)src;
    uint32_t len;
    uint16_t ndims;
    memcpy(&len, s, 4);
    memcpy(&ndims, s + 4, 2);
    if (ndims == 0) return 0;                                 /* delta.cpp:191, :637 */
    if (elem_bytes == 1) dec_8(kind, s + 6, len, (uint8_t*)dest, ndims);
    else dec_16(kind, (const uint16_t*)(s + 6), len, (uint16_t*)dest, ndims);
    return len;
}

/*
 * transforms_oracle.c -- CPU restatement of the reference's stand-alone transforms
 * (SURVEY.md 8f-2).  TEST INFRASTRUCTURE ONLY.  Parity: PINNED against the compiled
 * reference (oracle/_ref, ref_transform_*) by tests/test_transforms_cpu.py and the fixtures
 * in tests/golden/golden_transforms_v2.npz.
 *
 *   kind 0  delta         encode_delta_rowmajor        cpp/Compress/delta.cpp:35-121
 *                         decode_delta_rowmajor        delta.cpp:133-232,309-397
 *   kind 1  double delta  encode_doubledelta_rowmajor  delta.cpp:405-465,532-610
 *                         decode_doubledelta_rowmajor  delta.cpp:467-529,623-693
 *   kind 2  xff (FIRE)    encode_xff_rowmajor          cpp/Compress/predict.cpp:57-289
 *                         decode_xff_rowmajor          predict.cpp:302-517
 *
 * Container: 6-byte header {u32 len; u16 ndims} (format.h:65-86) = 6 elements at 8 bits,
 * 3 at 16, then `len` transformed elements.  Arithmetic wraps at the element width.
 *
 * What the vectorised reference code computes, stripped of its striping: per column c
 * (element index mod ndims), with the state starting at zero,
 *   delta:         y[r] = x[r] - x[r-1]
 *   double delta:  d[r] = x[r] - x[r-1];  y[r] = d[r] - d[r-1]
 * (the first row is "x - 0", i.e. copied; delta.cpp:107-111, :432-440).
 *
 * xff is the FIRE forecaster with NO packing (errors out), and its constants are not the
 * codec's (predict.cpp:62: learning shift 3 at 16 bits, prediction << 2): per column, per
 * 8-row block, with prev value / prev delta / counter starting at zero,
 *   8 bits   coef = int16((ctr >> 5) << 4), ctr int16 (predict.cpp:140-145)
 *            pred = byte 1 of the 16-bit product  P * coef, where P is the previous delta taken
 *                   UNSIGNED in even columns and SIGNED in odd ones (the vector code multiplies
 *                   "deltas & 0xff" for the even lanes, predict.cpp:163-168) -- restated as is
 *            grad = int8 wrapping sum over the block's odd rows of sign(err) * prev_delta
 *                   (_mm256_sign_epi8: -(-128) stays -128; predict.cpp:173-184);  ctr += grad >> 2
 *   16 bits  coef = int16((ctr >> 15) << 12), ctr int32 (predict.cpp:224-232)
 *            pred = int16(mulhi(prev_delta, coef) << 2) (predict.cpp:240-242)
 *            grad = int16 wrapping sum likewise; ctr += grad >> 2
 *   err = delta - pred.  Only the first `nblocks` blocks are forecast: nblocks = rows / 8 minus
 *   ceil(overrun / (8 ndims)) when overrun = V - ndims % V (V = 32 / elem size; V, not 0, for
 *   aligned ndims) exceeds len % (8 ndims) (predict.cpp:96-103) -- the vector stores spill that
 *   far.  What follows is plain delta against the previous row (predict.cpp:271-273); with
 *   nblocks == 0 the first row is copied (:266-270).
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

static int32_t xff_nblocks(uint32_t len, uint32_t D, int esz)
{
    const uint32_t V = 32u / (uint32_t)esz, blk = 8u * D;
    int32_t nblocks = (int32_t)((len / D) / 8u);
    const uint16_t overrun = (uint16_t)(V - (D % V));
    const uint32_t trailing = len % blk;
    if (overrun > trailing) {
        nblocks -= (int32_t)((overrun + blk - 1) / blk);
        if (nblocks < 0) nblocks = 0;
    }
    return nblocks;
}

/* one column's forecast of the previous delta */
static int pred8(int pd, int ctr, int odd_col)
{
    const int16_t coef = (int16_t)((int16_t)(ctr >> 5) << 4);
    const int p = odd_col ? (int)(int8_t)pd : (int)(uint8_t)pd;
    return (int)(uint8_t)(((uint16_t)(p * coef)) >> 8);
}
static int pred16(int pd, int32_t ctr)
{
    const int16_t coef = (int16_t)((uint16_t)(ctr >> 15) << 12);
    const int16_t hi = (int16_t)(((int32_t)(int16_t)pd * (int32_t)coef) >> 16);
    return (int)(int16_t)((uint16_t)hi << 2);
}

static void xff8(int decode, const uint8_t* in, uint32_t len, uint8_t* out, uint32_t D)
{
    const int32_t nblocks = xff_nblocks(len, D, 1);
    for (uint32_t c = 0; c < D && nblocks > 0; c++) {
        uint8_t pv = 0;
        int8_t pd = 0;
        int16_t ctr = 0;
        for (int32_t b = 0; b < nblocks; b++) {
            int8_t grad = 0;
            for (int i = 0; i < 8; i++) {
                const size_t at = ((size_t)b * 8 + i) * D + c;
                const int8_t pr = (int8_t)pred8(pd, ctr, c & 1);
                int8_t delta, err;
                uint8_t x;
                if (decode) { err = (int8_t)in[at]; delta = (int8_t)(err + pr); x = (uint8_t)(pv + delta); out[at] = x; }
                else { x = in[at]; delta = (int8_t)(x - pv); err = (int8_t)(delta - pr); out[at] = (uint8_t)err; }
                if (i & 1) grad = (int8_t)(grad + (err > 0 ? pd : err < 0 ? (int8_t)-pd : 0));
                pv = x;
                pd = delta;
            }
            ctr = (int16_t)(ctr + (grad >> 2));
        }
    }
    for (size_t i = (size_t)nblocks * 8 * D; i < len; i++) {
        if (decode) out[i] = (uint8_t)(in[i] + (i >= D ? out[i - D] : 0));
        else out[i] = (uint8_t)(in[i] - (i >= D ? in[i - D] : 0));
    }
}

static void xff16(int decode, const uint16_t* in, uint32_t len, uint16_t* out, uint32_t D)
{
    const int32_t nblocks = xff_nblocks(len, D, 2);
    for (uint32_t c = 0; c < D && nblocks > 0; c++) {
        uint16_t pv = 0;
        int16_t pd = 0;
        int32_t ctr = 0;
        for (int32_t b = 0; b < nblocks; b++) {
            int16_t grad = 0;
            for (int i = 0; i < 8; i++) {
                const size_t at = ((size_t)b * 8 + i) * D + c;
                const int16_t pr = (int16_t)pred16(pd, ctr);
                int16_t delta, err;
                uint16_t x;
                if (decode) { err = (int16_t)in[at]; delta = (int16_t)(err + pr); x = (uint16_t)(pv + delta); out[at] = x; }
                else { x = in[at]; delta = (int16_t)(x - pv); err = (int16_t)(delta - pr); out[at] = (uint16_t)err; }
                if (i & 1) grad = (int16_t)(grad + (err > 0 ? pd : err < 0 ? (int16_t)-pd : 0));
                pv = x;
                pd = delta;
            }
            ctr = (int32_t)((uint32_t)ctr + (uint32_t)(int32_t)(grad >> 2));
        }
    }
    for (size_t i = (size_t)nblocks * 8 * D; i < len; i++) {
        if (decode) out[i] = (uint16_t)(in[i] + (i >= D ? out[i - D] : 0));
        else out[i] = (uint16_t)(in[i] - (i >= D ? in[i - D] : 0));
    }
}

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
    if (kind == 2) {
        if (elem_bytes == 1) xff8(0, (const uint8_t*)src, len, d, ndims);
        else xff16(0, (const uint16_t*)src, len, (uint16_t*)d, ndims);
        return len + hdr;
    }
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
    if (kind == 2) {
        if (elem_bytes == 1) xff8(1, s + 6, len, (uint8_t*)dest, ndims);
        else xff16(1, (const uint16_t*)(s + 6), len, (uint16_t*)dest, ndims);
        return len;
    }
    if (elem_bytes == 1) dec_8(kind, s + 6, len, (uint8_t*)dest, ndims);
    else dec_16(kind, (const uint16_t*)(s + 6), len, (uint16_t*)dest, ndims);
    return len;
}

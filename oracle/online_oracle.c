/*
 * online_oracle.c -- CPU restatement of the reference's 2020 "online" u16 coders
 * (cpp/Compress/online.hpp:395-445, online.cpp).  TEST INFRASTRUCTURE ONLY: lives in liboracle.so,
 * pinned against the compiled reference where oracle/_ref exists (tests/test_online_cpu.py) and
 * against fixtures minted from it (tests/golden/golden_online_v1.npz).
 *
 * All three families are 1-D uint16 streams behind a 4-byte header {u32 len} (format.h:33,89-99:
 * write_metadata_simple1d, 2 elements).  Return values are in ELEMENTS like the reference's.
 *
 * dynamic delta (online.cpp:48-311).  Element 0 verbatim; then blocks of 8 elements (1 + 8b ...), each coded
 *   by whichever of two predictors has the smaller loss over the block's zigzagged errors:
 *     0  delta          err = x[i] - x[i-1]
 *     1  double delta   err = x[i] - (x[i-1] + (x[i-1] - x[i-2])),  x[-1] := x[0]   (online.hpp: _prev_diff starts at 0)
 *   both predictors see every TRUE value (the encoder trains both, the decoder jump()s the idle one), so a
 *   block's choice is a pure function of the input.  loss0 <= loss1 picks delta (:119).  Losses (:17-45):
 *     MaxAbs     the largest zigzagged error
 *     SumLogAbs  sum of (uint8_t)(16 - clz32(v)): clz is taken of the value promoted to 32 bits, so the term
 *                is 241 + floor(log2 v) for 1 <= v < 32768, 0 for v >= 32768 and, lzcnt giving 32 for 0 on
 *                the reference's required ISA (-mlzcnt), 240 for v == 0 -- restated as the compiled code behaves.
 *   The errors of full blocks are written zigzagged; the < 8 trailing elements are delta errors, NOT zigzagged
 *   (:149-155).  The choices (one bit per block, LSB first) follow the len elements; the container reserves
 *   ceil(ceil(len / 8) / 8) bytes for them, rounded up to an element (:253-287).
 * zigzag (online.cpp:314-351): every value as int16 through (x << 1) ^ (x >> 15).
 * sprintzpack (online.cpp:355-703): after the header, ceil(len / 8) header nibbles (two blocks a byte, low nibble
 *   first; rounded up to an element), then per FULL block of 8 values nbits bytes: the 8 values' low nbits bits,
 *   LSB first (two u64 halves of 4 values, :430-452 -- contiguous because 4 nbits bits is a whole number of
 *   nibbles and 8 nbits bits a whole number of bytes); nbits = bit length of the OR, 15 counted as 16
 *   (bitpack.h:273-287), nibble = nbits - (nbits == 16).  Values optionally zigzagged first.  The < 8 trailing
 *   values are stored raw (2 bytes each) right after the last payload byte (any byte alignment).
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>

static uint16_t zz16(int16_t x) { return (uint16_t)(((uint16_t)x << 1) ^ (uint16_t)(x >> 15)); }
static int16_t unzz16(uint16_t x) { return (int16_t)((x >> 1) ^ (uint16_t)(-(int16_t)(x & 1))); }
static int bitlen32(uint32_t v) { int n = 0; while (v) { n++; v >>= 1; } return n; }

static int32_t loss_of(const uint16_t* z, int alt)
{
    if (alt) {                                            /* MaxAbs (:26-35) */
        int32_t m = z[0];
        for (int i = 1; i < 8; i++) if (z[i] > m) m = z[i];
        return m;
    }
    int32_t s = 0;                                        /* SumLogAbs (:36-43), as compiled with lzcnt */
    for (int i = 0; i < 8; i++) s += (uint8_t)(16 - (32 - bitlen32(z[i])));
    return s;
}

/* kind: 0 dynamic delta (SumLogAbs), 1 dynamic delta (MaxAbs), 2 zigzag, 3 sprintzpack, 4 sprintzpack + zigzag.
 * dest must hold online_oracle_bound(kind, len) bytes.  *nbytes = bytes of the container (exact). */
size_t online_oracle_bound(int kind, uint32_t len)
{
    (void)kind;
    return 4 + (size_t)len * 2 + ((size_t)len + 7) / 8 + 64;
}

int64_t online_oracle_pack(int kind, const uint16_t* x, uint32_t len, uint8_t* dest, size_t* nbytes)
{
    memcpy(dest, &len, 4);
    uint8_t* p = dest + 4;
    if (kind == 2) {
        for (uint32_t i = 0; i < len; i++) { const uint16_t z = zz16((int16_t)x[i]); memcpy(p + 2 * (size_t)i, &z, 2); }
        if (nbytes) *nbytes = 4 + (size_t)len * 2;
        return 2 + (int64_t)len;
    }
    if (kind <= 1) {
        const uint32_t cbytes = (((len + 7) / 8) + 7) / 8, celems = (cbytes + 1) / 2;
        uint8_t* choices = p + 2 * (size_t)len;
        memset(choices, 0, (size_t)celems * 2);
        if (len >= 1) memcpy(p, &x[0], 2);
        if (len >= 2) {
            const uint32_t n = len - 1, nblocks = n / 8;
            for (uint32_t b = 0; b < nblocks; b++) {
                uint16_t z0[8], z1[8];
                for (int i = 0; i < 8; i++) {
                    const uint32_t at = 1 + 8 * b + (uint32_t)i;
                    const uint16_t p1 = x[at - 1], p2 = at >= 2 ? x[at - 2] : x[0];
                    z0[i] = zz16((int16_t)(uint16_t)(x[at] - p1));
                    z1[i] = zz16((int16_t)(uint16_t)(x[at] - (uint16_t)(p1 + (uint16_t)(p1 - p2))));
                }
                const int choice = loss_of(z0, kind) <= loss_of(z1, kind) ? 0 : 1;
                memcpy(p + 2 * (size_t)(1 + 8 * b), choice ? z1 : z0, 16);
                choices[b / 8] |= (uint8_t)(choice << (b % 8));
            }
            for (uint32_t at = 1 + 8 * nblocks; at < len; at++) {
                const uint16_t e = (uint16_t)(x[at] - x[at - 1]);
                memcpy(p + 2 * (size_t)at, &e, 2);
            }
        }
        if (nbytes) *nbytes = 4 + (size_t)len * 2 + (size_t)celems * 2;
        return 2 + (int64_t)len + celems;
    }
    /* sprintzpack */
    const int zig = kind == 4;
    const uint32_t nblocks_up = (len + 7) / 8, hbytes = (nblocks_up * 4 + 7) / 8, helems = (hbytes + 1) / 2;
    uint8_t* hdr = p;
    uint8_t* out = p + 2 * (size_t)helems;
    memset(hdr, 0, (size_t)helems * 2);
    const uint32_t nblocks = len / 8;
    size_t pos = 0;
    for (uint32_t b = 0; b < nblocks; b++) {
        uint16_t v[8];
        uint32_t all = 0;
        for (int i = 0; i < 8; i++) { v[i] = zig ? zz16((int16_t)x[8 * b + i]) : x[8 * b + i]; all |= v[i]; }
        int nb = bitlen32(all);
        if (nb == 15) nb = 16;
        const int nib = nb - (nb == 16);
        hdr[b / 2] |= (uint8_t)(nib << (4 * (b % 2)));
        uint64_t acc = 0;
        int have = 0;
        for (int i = 0; i < 8; i++) {
            acc |= (uint64_t)(v[i] & (uint16_t)((1u << nb) - 1u)) << have;
            have += nb;
            while (have >= 8) { out[pos++] = (uint8_t)acc; acc >>= 8; have -= 8; }
        }
    }
    for (uint32_t at = 8 * nblocks; at < len; at++) { memcpy(out + pos, &x[at], 2); pos += 2; }
    const size_t total = 4 + (size_t)helems * 2 + pos;
    if (total & 1) dest[total] = 0;
    if (nbytes) *nbytes = total;
    return 2 + (int64_t)helems + (int64_t)((pos + 1) / 2);
}

/* -> elements decoded (= the header's len); dest must hold len elements */
int64_t online_oracle_unpack(int kind, const uint8_t* src, uint16_t* out)
{
    uint32_t len;
    memcpy(&len, src, 4);
    const uint8_t* p = src + 4;
    if (kind == 2) {
        for (uint32_t i = 0; i < len; i++) { uint16_t z; memcpy(&z, p + 2 * (size_t)i, 2); out[i] = (uint16_t)unzz16(z); }
        return len;
    }
    if (kind <= 1) {
        const uint8_t* choices = p + 2 * (size_t)len;
        if (len == 0) return 0;
        memcpy(&out[0], p, 2);
        const uint32_t n = len - 1, nblocks = n / 8;
        uint16_t prev = out[0];
        int16_t diff = 0;                                  /* the true last difference (what jump() restores) */
        for (uint32_t b = 0; b < nblocks; b++) {
            const int choice = (choices[b / 8] >> (b % 8)) & 1;
            for (int i = 0; i < 8; i++) {
                uint16_t z;
                memcpy(&z, p + 2 * (size_t)(1 + 8 * b + (uint32_t)i), 2);
                const int16_t e = unzz16(z);
                const uint16_t pred = choice ? (uint16_t)(prev + (uint16_t)diff) : prev;
                const uint16_t v = (uint16_t)(pred + (uint16_t)e);
                diff = (int16_t)(uint16_t)(v - prev);
                prev = v;
                out[1 + 8 * b + (uint32_t)i] = v;
            }
        }
        for (uint32_t at = 1 + 8 * nblocks; at < len; at++) {
            uint16_t e;
            memcpy(&e, p + 2 * (size_t)at, 2);
            prev = (uint16_t)(prev + e);
            out[at] = prev;
        }
        return len;
    }
    const int zig = kind == 4;
    const uint32_t nblocks_up = (len + 7) / 8, hbytes = (nblocks_up * 4 + 7) / 8, helems = (hbytes + 1) / 2;
    const uint8_t* hdr = p;
    const uint8_t* in = p + 2 * (size_t)helems;
    const uint32_t nblocks = len / 8;
    size_t pos = 0;
    for (uint32_t b = 0; b < nblocks; b++) {
        int nb = (hdr[b / 2] >> (4 * (b % 2))) & 15;
        nb += nb == 15;
        uint64_t acc = 0;
        int have = 0;
        for (int i = 0; i < 8; i++) {
            while (have < nb) { acc |= (uint64_t)in[pos++] << have; have += 8; }
            const uint16_t v = (uint16_t)(acc & ((1ull << nb) - 1ull));
            acc >>= nb;
            have -= nb;
            out[8 * b + (uint32_t)i] = zig ? (uint16_t)unzz16(v) : v;
        }
    }
    for (uint32_t at = 8 * nblocks; at < len; at++) { memcpy(&out[at], in + pos, 2); pos += 2; }
    return len;
}

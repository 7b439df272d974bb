/*
 * huf0_oracle.c -- CPU restatement of the Huff0 BLOCK DECODER (Yann Collet's Huff0 /
 * FiniteStateEntropy, the entropy coder the paper applies after bit-packing:
 * communicate/ubicomp/method.tex:293-297).  TEST INFRASTRUCTURE ONLY.
 *
 * The coder is a third-party dependency that dblalock/sprintz neither vendors nor pins
 * (no submodule, no manifest, no call site in cpp/).  What is restated here is
 * the published format as implemented in zstd 1.4.8's lib/common/{entropy_common,fse_decompress}.c
 * and lib/decompress/huf_decompress.c (HUF_decompress = tree description + 4 interleaved
 * streams), the version the system libzstd.so.1 exports.  Parity: PINNED against that library
 * -- tests/test_huf0_cpu.py decodes blocks written by its HUF_compress and compares with its
 * HUF_decompress, and tests/golden/golden_huf0_v1.npz holds blocks it wrote (minted by
 * oracle/gen_golden_huf0.py).
 *
 * A block, as HUF_compress writes it:
 *   tree description   1 byte h.  h >= 128: h - 127 weights follow, 4 bits each, high nibble
 *                      first.  h < 128: h bytes of FSE-compressed weights follow (table log <= 6).
 *                      weight w > 0 means a code of tableLog + 1 - w bits; the LAST symbol's
 *                      weight is implied (the sum of 2^(w-1) must reach a power of two).
 *   jump table         3 x u16 LE: byte sizes of streams 1..3
 *   4 bit streams      each encodes ceil(n / 4) symbols (the last one what is left) written
 *                      LSB-first, last symbol FIRST, closed by a 1 bit; the decoder reads from
 *                      the top of the last byte down, tableLog bits of look-ahead at a time.
 * Codes are canonical: per code length ascending symbols, the longest codes lowest
 * (HUF_readDTableX1's fill order).
 * HUF_decompress conventions kept: src_size == dst_size means stored, src_size == 1 means one
 * repeated byte.
 */
#include <stdint.h>
#include <string.h>

#define HUF0_TABLELOG_MAX 12
#define HUF0_ERR (-1)

static int highbit(uint32_t v) { int r = -1; while (v) { r++; v >>= 1; } return r; }

/* ---- the backward bit reader (bitstream.h BIT_DStream_t), as a cursor P = unread bits */
typedef struct { const uint8_t* p; int64_t P; } back_t;
static int back_init(back_t* b, const uint8_t* p, size_t n)
{
    if (n < 1 || p[n - 1] == 0) return HUF0_ERR;
    b->p = p;
    b->P = 8 * (int64_t)(n - 1) + highbit(p[n - 1]);
    return 0;
}
/* the nb bits below the cursor, most significant first; positions before the start read 0 */
static uint32_t back_look(const back_t* b, int nb)
{
    uint32_t v = 0;
    for (int k = 0; k < nb; k++) {
        const int64_t pos = b->P - 1 - k;
        v = (v << 1) | (pos >= 0 ? (uint32_t)((b->p[pos >> 3] >> (pos & 7)) & 1) : 0u);
    }
    return v;
}
static uint32_t back_read(back_t* b, int nb) { const uint32_t v = back_look(b, nb); b->P -= nb; return v; }

/* ---- FSE_readNCount (entropy_common.c): a forward LSB-first reader over the header */
static uint32_t fwd32(const uint8_t* p, size_t n, uint64_t bp)
{
    uint32_t v = 0;
    for (int k = 0; k < 32; k++) {
        const uint64_t pos = bp + (uint64_t)k;
        if ((pos >> 3) < n) v |= (uint32_t)((p[pos >> 3] >> (pos & 7)) & 1) << k;
    }
    return v;
}
static int64_t fse_read_ncount(int16_t* norm, unsigned* max_sv, unsigned* table_log, const uint8_t* p, size_t n)
{
    uint64_t bp = 0;
    int nb = (int)(fwd32(p, n, bp) & 0xf) + 5;
    if (nb > 15) return HUF0_ERR;
    bp += 4;
    *table_log = (unsigned)nb;
    int remaining = (1 << nb) + 1, threshold = 1 << nb;
    nb++;
    unsigned charnum = 0;
    int previous0 = 0;
    memset(norm, 0, 256 * sizeof(int16_t));
    while (remaining > 1 && charnum <= *max_sv) {
        if (previous0) {
            unsigned n0 = charnum;
            while ((fwd32(p, n, bp) & 0xffff) == 0xffff) { n0 += 24; bp += 16; if (bp > 8 * n + 32) return HUF0_ERR; }
            while ((fwd32(p, n, bp) & 3) == 3) { n0 += 3; bp += 2; }
            n0 += fwd32(p, n, bp) & 3;
            bp += 2;
            if (n0 > *max_sv) return HUF0_ERR;
            while (charnum < n0) norm[charnum++] = 0;
        }
        {
            const uint32_t bits = fwd32(p, n, bp);
            const int max = (2 * threshold - 1) - remaining;
            int count;
            if ((int)(bits & (uint32_t)(threshold - 1)) < max) {
                count = (int)(bits & (uint32_t)(threshold - 1));
                bp += (uint64_t)(nb - 1);
            } else {
                count = (int)(bits & (uint32_t)(2 * threshold - 1));
                if (count >= threshold) count -= max;
                bp += (uint64_t)nb;
            }
            count--;
            remaining -= count < 0 ? -count : count;
            if (charnum > 255) return HUF0_ERR;
            norm[charnum++] = (int16_t)count;
            previous0 = !count;
            while (remaining < threshold) { nb--; threshold >>= 1; }
        }
    }
    if (remaining != 1) return HUF0_ERR;
    if (bp > 8 * (uint64_t)n) return HUF0_ERR;
    *max_sv = charnum - 1;
    return (int64_t)((bp + 7) >> 3);
}

/* ---- FSE_decompress_wksp (fse_decompress.c) for the weights: table log <= 6 */
static int64_t fse_decompress(uint8_t* dst, size_t cap, const uint8_t* src, size_t n, unsigned max_log)
{
    int16_t norm[256];
    unsigned max_sv = 255, tl = 0;
    const int64_t hl = fse_read_ncount(norm, &max_sv, &tl, src, n);
    if (hl < 0 || tl > max_log || (size_t)hl >= n + 1) return HUF0_ERR;
    struct { uint16_t new_state; uint8_t symbol, nbits; } dt[1 << 6];
    uint16_t next[256];
    const unsigned size = 1u << tl;
    unsigned high = size - 1;
    for (unsigned s = 0; s <= max_sv; s++) {
        if (norm[s] == -1) { dt[high--].symbol = (uint8_t)s; next[s] = 1; }
        else next[s] = (uint16_t)norm[s];
    }
    {
        const unsigned mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
        unsigned pos = 0;
        for (unsigned s = 0; s <= max_sv; s++)
            for (int i = 0; i < norm[s]; i++) {
                dt[pos].symbol = (uint8_t)s;
                pos = (pos + step) & mask;
                while (pos > high) pos = (pos + step) & mask;
            }
        if (pos != 0) return HUF0_ERR;
    }
    for (unsigned u = 0; u < size; u++) {
        const unsigned ns = next[dt[u].symbol]++;
        if (ns == 0 || ns >= 2 * size) return HUF0_ERR;
        dt[u].nbits = (uint8_t)(tl - (unsigned)highbit(ns));
        dt[u].new_state = (uint16_t)((ns << dt[u].nbits) - size);
    }
    back_t b;
    if (back_init(&b, src + hl, n - (size_t)hl)) return HUF0_ERR;
    unsigned s1 = back_read(&b, (int)tl), s2 = back_read(&b, (int)tl);
    size_t o = 0;
    for (;;) {   /* two interleaved states; the stream ends by running dry, the other state holds the last symbol */
        if (o + 2 > cap) return HUF0_ERR;
        dst[o++] = dt[s1].symbol;
        s1 = dt[s1].new_state + back_read(&b, dt[s1].nbits);
        if (b.P < 0) { dst[o++] = dt[s2].symbol; break; }
        if (o + 2 > cap) return HUF0_ERR;
        dst[o++] = dt[s2].symbol;
        s2 = dt[s2].new_state + back_read(&b, dt[s2].nbits);
        if (b.P < 0) { dst[o++] = dt[s1].symbol; break; }
    }
    return (int64_t)o;
}

/* ---- HUF_readStats (entropy_common.c): weights[0..nsym), table log; returns header bytes */
int64_t oracle_huf0_read_stats(uint8_t* weights, unsigned* nsym, unsigned* table_log, const uint8_t* src, size_t n)
{
    if (n < 1) return HUF0_ERR;
    size_t isize = src[0], osize;
    if (isize >= 128) {
        osize = isize - 127;
        isize = (osize + 1) / 2;
        if (isize + 1 > n || osize >= 256) return HUF0_ERR;
        for (size_t k = 0; k < osize; k += 2) {
            weights[k] = src[1 + k / 2] >> 4;
            weights[k + 1] = src[1 + k / 2] & 15;
        }
    } else {
        if (isize + 1 > n) return HUF0_ERR;
        const int64_t r = fse_decompress(weights, 255, src + 1, isize, 6);
        if (r < 0) return HUF0_ERR;
        osize = (size_t)r;
    }
    uint32_t rank[HUF0_TABLELOG_MAX + 1] = {0}, total = 0;
    for (size_t k = 0; k < osize; k++) {
        if (weights[k] >= HUF0_TABLELOG_MAX) return HUF0_ERR;
        rank[weights[k]]++;
        total += (1u << weights[k]) >> 1;
    }
    if (total == 0) return HUF0_ERR;
    const unsigned tl = (unsigned)highbit(total) + 1;
    if (tl > HUF0_TABLELOG_MAX) return HUF0_ERR;
    const uint32_t rest = (1u << tl) - total;
    if ((1u << highbit(rest)) != rest) return HUF0_ERR;
    weights[osize] = (uint8_t)(highbit(rest) + 1);
    rank[weights[osize]]++;
    if (rank[1] < 2 || (rank[1] & 1)) return HUF0_ERR;
    *nsym = (unsigned)osize + 1;
    *table_log = tl;
    return (int64_t)isize + 1;
}

/* ---- HUF_decompress (huf_decompress.c): single-symbol table, 4 streams */
int64_t oracle_huf0_decompress(uint8_t* dst, size_t dst_size, const uint8_t* src, size_t src_size)
{
    if (dst_size == 0 || src_size > dst_size || src_size == 0) return HUF0_ERR;
    if (src_size == dst_size) { memcpy(dst, src, dst_size); return (int64_t)dst_size; }
    if (src_size == 1) { memset(dst, src[0], dst_size); return (int64_t)dst_size; }
    uint8_t weights[256];
    unsigned nsym = 0, tl = 0;
    const int64_t hl = oracle_huf0_read_stats(weights, &nsym, &tl, src, src_size);
    if (hl < 0 || (size_t)hl >= src_size) return HUF0_ERR;
    /* HUF_readDTableX1: per weight ascending symbols, weight 1 (the longest codes) lowest */
    static _Thread_local struct { uint8_t byte, nbits; } dt[1 << HUF0_TABLELOG_MAX];
    uint32_t start[HUF0_TABLELOG_MAX + 2] = {0}, cnt[HUF0_TABLELOG_MAX + 2] = {0};
    for (unsigned s = 0; s < nsym; s++) cnt[weights[s]]++;
    for (unsigned w = 1, at = 0; w <= tl; w++) { start[w] = at; at += cnt[w] << (w - 1); }
    for (unsigned s = 0; s < nsym; s++) {
        const unsigned w = weights[s];
        if (!w) continue;
        const uint32_t len = (1u << w) >> 1;
        for (uint32_t u = start[w]; u < start[w] + len; u++) { dt[u].byte = (uint8_t)s; dt[u].nbits = (uint8_t)(tl + 1 - w); }
        start[w] += len;
    }
    const uint8_t* ip = src + hl;
    const size_t n = src_size - (size_t)hl;
    if (n < 10) return HUF0_ERR;
    const size_t l1 = ip[0] | (ip[1] << 8), l2 = ip[2] | (ip[3] << 8), l3 = ip[4] | (ip[5] << 8);
    if (6 + l1 + l2 + l3 > n) return HUF0_ERR;
    const size_t l4 = n - 6 - l1 - l2 - l3;
    const size_t lens[4] = {l1, l2, l3, l4};
    const size_t seg = (dst_size + 3) / 4;
    const uint8_t* sp = ip + 6;
    for (int k = 0; k < 4; k++) {
        const size_t o0 = seg * (size_t)k < dst_size ? seg * (size_t)k : dst_size;
        const size_t o1 = k == 3 ? dst_size : (o0 + seg < dst_size ? o0 + seg : dst_size);
        back_t b;
        if (back_init(&b, sp, lens[k])) return HUF0_ERR;
        for (size_t o = o0; o < o1; o++) {
            const uint32_t idx = back_look(&b, (int)tl);
            dst[o] = dt[idx].byte;
            b.P -= dt[idx].nbits;
        }
        if (b.P != 0) return HUF0_ERR;                    /* every stream must end exactly (BIT_endOfDStream) */
        sp += lens[k];
    }
    return (int64_t)dst_size;
}

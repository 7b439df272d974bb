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

/* =====================================================================================
 * WRITER.  Not a restatement of HUF_compress: the format leaves the encoder free (which code
 * lengths, how the weight statistics are normalised), so this is the specification of the blocks
 * sprintz_mi355x_huf0_compress_batch writes -- every one of them a block the library's
 * HUF_decompress accepts (tests/test_huf0_cpu.py feeds them to it) -- and the GPU kernels are
 * byte-exact with it.
 *   * code lengths: one table per segment of 64 chunks (huf_oracle_lengths over the segment's
 *     histogram, <= 11 bits), written into EVERY chunk's block; a table that is not Kraft-complete
 *     or has fewer than two symbols makes the segment's chunks stored.
 *   * codes: Huff0's canonical order (HUF_buildCTable: per length ascending symbols, values
 *     descending with the length).
 *   * tree description: the weights FSE-coded with table log 6 (counts scaled to 64, every present
 *     weight >= 1, the remainder given to / taken from the most frequent one; FSE_writeNCount's bit
 *     layout; states chosen by search in the decoder's table, the last symbol of each of the two
 *     state chains on that symbol's smallest state as FSE_initCState2 does) when that is shorter
 *     than the 4-bit form or the 4-bit form does not exist (> 128 weights).
 *   * a chunk is stored when it is shorter than 12 bytes, a stream would not fit the 16-bit jump
 *     table, or the block would not be smaller; a chunk of one repeated byte is that byte.
 */
void huf_oracle_lengths(const uint32_t counts[256], uint8_t lens[256]);

typedef struct { uint8_t* p; uint64_t acc; int nbits; size_t n; } bitw_t;
static void bw_add(bitw_t* b, uint32_t v, int nb)
{
    b->acc |= (uint64_t)v << b->nbits;
    b->nbits += nb;
    while (b->nbits >= 8) { b->p[b->n++] = (uint8_t)b->acc; b->acc >>= 8; b->nbits -= 8; }
}
static size_t bw_close(bitw_t* b, int end_mark)
{
    if (end_mark) bw_add(b, 1, 1);
    if (b->nbits > 0) { b->p[b->n++] = (uint8_t)b->acc; b->acc = 0; b->nbits = 0; }
    return b->n;
}

/* FSE-coded weights: out[0..) = NCount + state stream; returns its size, 0 if not applicable */
static size_t fse_write_weights(const uint8_t* w, unsigned n, uint8_t* out)
{
    unsigned count[16] = {0}, maxw = 0;
    if (n < 2) return 0;
    for (unsigned k = 0; k < n; k++) { count[w[k]]++; if (w[k] > maxw) maxw = w[k]; }
    unsigned present = 0, top = 0;
    for (unsigned s = 0; s <= maxw; s++) { present += count[s] != 0; if (count[s] > count[top]) top = s; }
    if (present < 2) return 0;                               /* one repeated weight: FSE cannot code it */
    const unsigned tl = 6, size = 64;
    int norm[16] = {0}, sum = 0;
    for (unsigned s = 0; s <= maxw; s++) if (count[s]) { norm[s] = (int)((uint64_t)count[s] * size / n); if (norm[s] < 1) norm[s] = 1; sum += norm[s]; }
    norm[top] += (int)size - sum;                            /* the most frequent weight absorbs the rounding */
    if (norm[top] < 1) return 0;
    /* FSE_writeNCount */
    bitw_t b = {out, 0, 0, 0};
    bw_add(&b, tl - 5, 4);
    {
        int remaining = (int)size + 1, threshold = (int)size, nb = (int)tl + 1, previous0 = 0;
        unsigned sym = 0;
        while (sym <= maxw && remaining > 1) {
            if (previous0) {
                unsigned start = sym;
                while (sym <= maxw && !norm[sym]) sym++;
                if (sym > maxw) return 0;
                while (sym >= start + 24) { start += 24; bw_add(&b, 0xffff, 16); }
                while (sym >= start + 3) { start += 3; bw_add(&b, 3, 2); }
                bw_add(&b, sym - start, 2);
            }
            int c = norm[sym++];
            const int max = (2 * threshold - 1) - remaining;
            remaining -= c;
            c++;
            if (c >= threshold) c += max;
            bw_add(&b, (uint32_t)c, nb - (c < max));
            previous0 = c == 1;
            if (remaining < 1) return 0;
            while (remaining < threshold) { nb--; threshold >>= 1; }
        }
        if (remaining != 1) return 0;
    }
    const size_t hl = bw_close(&b, 0);
    /* the decoder's table (FSE_buildDTable), then the states by search */
    uint8_t tsym[64], tnb[64];
    uint16_t tnew[64], next[16];
    {
        const unsigned mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
        unsigned pos = 0;
        for (unsigned s = 0; s <= maxw; s++) {
            next[s] = (uint16_t)norm[s];
            for (int i = 0; i < norm[s]; i++) { tsym[pos] = (uint8_t)s; pos = (pos + step) & mask; }
        }
        for (unsigned u = 0; u < size; u++) {
            const unsigned ns = next[tsym[u]]++;
            tnb[u] = (uint8_t)(tl - (unsigned)highbit(ns));
            tnew[u] = (uint16_t)((ns << tnb[u]) - size);
        }
    }
    /* D[i] = decoder state when symbol i is emitted; chains i, i+2, ...; fields in reverse read order */
    unsigned D[256];
    for (int i = (int)n - 1; i >= 0; i--) {
        int found = -1;
        for (unsigned u = 0; u < size && found < 0; u++) {
            if (tsym[u] != w[i]) continue;
            if (i + 2 >= (int)n) found = (int)u;                           /* smallest state of the symbol: table order */
            else if (D[i + 2] >= tnew[u] && D[i + 2] < (unsigned)tnew[u] + (1u << tnb[u])) found = (int)u;
        }
        if (found < 0) return 0;
        D[i] = (unsigned)found;
    }
    bitw_t sb = {out + hl, 0, 0, 0};
    for (int i = (int)n - 3; i >= 0; i--) bw_add(&sb, D[i + 2] - tnew[D[i]], tnb[D[i]]);
    bw_add(&sb, D[1], (int)tl);
    bw_add(&sb, D[0], (int)tl);
    return hl + bw_close(&sb, 1);
}

/* tree description for code lengths lens[]: hdr[0..return); 0 when the table cannot be written */
size_t oracle_huf0_write_header(const uint8_t lens[256], uint8_t* hdr, unsigned* tl_out)
{
    unsigned tl = 0, nz = 0;
    int max_sym = -1;
    for (int s = 0; s < 256; s++) if (lens[s]) { nz++; max_sym = s; if (lens[s] > tl) tl = lens[s]; }
    if (nz < 2 || tl > 11) return 0;
    uint32_t kraft = 0;
    for (int s = 0; s < 256; s++) if (lens[s]) kraft += 1u << (tl - lens[s]);
    if (kraft != (1u << tl)) return 0;
    uint8_t w[256];
    const unsigned nw = (unsigned)max_sym;                   /* the last symbol's weight is implied */
    for (unsigned s = 0; s < nw; s++) w[s] = lens[s] ? (uint8_t)(tl + 1 - lens[s]) : 0;
    *tl_out = tl;
    uint8_t f[300];
    const size_t fs = nw >= 2 ? fse_write_weights(w, nw, f) : 0;
    const size_t raw = nw <= 128 ? 1 + (nw + 1) / 2 : 0;
    if (fs > 1 && fs < 128 && (raw == 0 || fs + 1 < raw)) {
        hdr[0] = (uint8_t)fs;
        memcpy(hdr + 1, f, fs);
        return fs + 1;
    }
    if (raw == 0 || nw == 0) return 0;
    hdr[0] = (uint8_t)(127 + nw);
    for (unsigned k = 0; k < nw; k += 2) hdr[1 + k / 2] = (uint8_t)((w[k] << 4) | (k + 1 < nw ? w[k + 1] : 0));
    return raw;
}

/* Huff0's canonical code values (HUF_buildCTable) */
static void huf0_codes(const uint8_t lens[256], unsigned tl, uint16_t val[256])
{
    uint32_t per[16] = {0}, start[16] = {0};
    for (int s = 0; s < 256; s++) per[lens[s]]++;
    uint32_t min = 0;
    for (unsigned l = tl; l > 0; l--) { start[l] = min; min += per[l]; min >>= 1; }
    for (int s = 0; s < 256; s++) val[s] = lens[s] ? (uint16_t)start[lens[s]]++ : 0;
}

/* container (dense, offsets, sizes) -> Huff0 blocks, one per chunk, byte-dense; returns total bytes */
uint64_t oracle_huf0_compress_batch(const uint8_t* dense, const uint64_t* offsets, const uint32_t* sizes, uint64_t nchunks,
                                    uint8_t* out, uint64_t* out_offsets)
{
    uint64_t at = 0;
    for (uint64_t c0 = 0; c0 < nchunks; c0 += 64) {
        const uint64_t c1 = c0 + 64 < nchunks ? c0 + 64 : nchunks;
        uint32_t hist[256] = {0};
        for (uint64_t c = c0; c < c1; c++) for (uint32_t k = 0; k < sizes[c]; k++) hist[dense[offsets[c] + k]]++;
        uint8_t lens[256], hdr[160];
        uint16_t val[256];
        unsigned tl = 0;
        huf_oracle_lengths(hist, lens);
        const size_t hlen = oracle_huf0_write_header(lens, hdr, &tl);
        if (hlen) huf0_codes(lens, tl, val);
        for (uint64_t c = c0; c < c1; c++) {
            const uint8_t* s = dense + offsets[c];
            const uint32_t n = sizes[c];
            out_offsets[c] = at;
            if (n == 0) continue;
            int same = 1;
            for (uint32_t k = 1; k < n; k++) if (s[k] != s[0]) { same = 0; break; }
            if (same) { out[at++] = s[0]; continue; }        /* n == 1 included: stored and repeated coincide */
            const uint32_t seg = (n + 3) / 4;
            uint64_t bits[4] = {0, 0, 0, 0}, bytes[4], total = hlen + 6;
            int ok = hlen != 0 && n >= 12;
            if (ok) {
                for (uint32_t k = 0; k < n; k++) bits[k / seg] += lens[s[k]];
                for (int j = 0; j < 4; j++) { bytes[j] = (bits[j] + 1 + 7) / 8; total += bytes[j]; if (j < 3 && bytes[j] > 65535) ok = 0; }
                if (total >= n) ok = 0;
            }
            if (!ok) { memcpy(out + at, s, n); at += n; continue; }
            uint8_t* o = out + at;
            memcpy(o, hdr, hlen);
            o += hlen;
            for (int j = 0; j < 3; j++) { o[2 * j] = (uint8_t)bytes[j]; o[2 * j + 1] = (uint8_t)(bytes[j] >> 8); }
            o += 6;
            for (int j = 0; j < 4; j++) {
                const uint32_t k0 = seg * (uint32_t)j, k1 = j == 3 ? n : k0 + seg;
                bitw_t b = {o, 0, 0, 0};
                for (uint32_t k = k1; k > k0; k--) bw_add(&b, val[s[k - 1]], lens[s[k - 1]]);   /* last symbol first */
                bw_close(&b, 1);
                o += bytes[j];
            }
            at += total;
        }
    }
    out_offsets[nchunks] = at;
    return at;
}

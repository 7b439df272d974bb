/*
 * huf_oracle.c -- CPU restatement of THIS repository's optional Huffman stage.
 * TEST INFRASTRUCTURE ONLY (see sprintz_oracle.h for the rules).
 *
 * Parity status: UNPINNED against the reference.  dblalock/sprintz contains no
 * Huffman coder: the paper entropy-codes the bit-packed bytes of each block with
 * Huff0 (communicate/ubicomp/method.tex:293-297), which lives in the author's
 * lzbench fork and is neither vendored nor version-pinned in the reference tree
 * (cpp/Compress/entropy.cpp:32-44 has empty tables).  What this file
 * pins is our own container format, so that the HIP kernels can be checked bit for
 * bit; the compression it buys is cross-checked against the system's Huff0
 * (libzstd HUF_compress) in tests/test_huf_cpu.py where that library exists.
 *
 * Format ("SPZH"), applied AFTER bit-packing, bytes as symbols (as the paper does):
 *   segment   = 64 consecutive chunks share one code table
 *   table     = 128 bytes per segment: 256 nibbles, nibble s = code length of byte
 *               value s (0 = absent, 1..11); low nibble = even s
 *   record c  = at huf_offsets[c] (4-byte aligned):
 *                 u32 hdr      bits 0..30 = n (bytes of the Sprintz stream), bit 31 = stored
 *                 stored:      n raw bytes
 *                 else:        u16 sz[3], u16 0  (encoded bytes of sub-streams 0..2)
 *                              4 sub-streams, byte aligned, concatenated; sub-stream j
 *                              holds symbols [j*q, min((j+1)*q, n)), q = ceil(n/4)
 *   bit order = LSB first; a symbol contributes its canonical code bit-REVERSED
 *               (so the decoder indexes a table with the low 11 bits)
 * Code lengths: Huffman (ties: lower count first, then lower symbol; a leaf before
 * an internal node of equal weight), limited to 11 bits by lengthening the deepest
 * codes, then any Kraft slack is given back to the most frequent symbols.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define HUF_LMAX 11
#define HUF_SEG 64

/* ------------------------------------------------------------ code lengths */

typedef struct { uint32_t cnt; uint32_t sym; } huf_key;

static int key_cmp(const void* a, const void* b)
{
    const huf_key* x = (const huf_key*)a; const huf_key* y = (const huf_key*)b;
    if (x->cnt != y->cnt) return x->cnt < y->cnt ? -1 : 1;
    return x->sym < y->sym ? -1 : (x->sym > y->sym);
}

void huf_oracle_lengths(const uint32_t counts[256], uint8_t lens[256])
{
    huf_key leaf[256];
    int nz = 0;
    memset(lens, 0, 256);
    for (int s = 0; s < 256; s++) if (counts[s]) { leaf[nz].cnt = counts[s]; leaf[nz].sym = (uint32_t)s; nz++; }
    if (nz == 0) return;
    if (nz == 1) { lens[leaf[0].sym] = 1; return; }
    qsort(leaf, (size_t)nz, sizeof(huf_key), key_cmp);

    /* two-queue Huffman: nodes 0..nz-1 leaves (sorted), nz..2nz-2 internal (creation order) */
    uint64_t w[511];
    int parent[511];
    for (int i = 0; i < nz; i++) w[i] = leaf[i].cnt;
    int ql = 0, qi = nz, next = nz;               /* heads of the leaf / internal queues */
    for (int m = 0; m < nz - 1; m++) {
        int pick[2];
        for (int t = 0; t < 2; t++) {
            const int has_l = ql < nz, has_i = qi < next;
            if (has_l && (!has_i || w[ql] <= w[qi])) pick[t] = ql++;    /* leaf first on ties */
            else pick[t] = qi++;
        }
        w[next] = w[pick[0]] + w[pick[1]];
        parent[pick[0]] = next;
        parent[pick[1]] = next;
        next++;
    }
    int depth[511];
    depth[next - 1] = 0;
    for (int i = next - 2; i >= 0; i--) depth[i] = depth[parent[i]] + 1;

    /* limit to HUF_LMAX, then repair / refill the Kraft sum (unit = 2^-HUF_LMAX) */
    uint32_t kraft = 0;
    for (int i = 0; i < nz; i++) {
        int l = depth[i] > HUF_LMAX ? HUF_LMAX : depth[i];
        lens[leaf[i].sym] = (uint8_t)l;
        kraft += 1u << (HUF_LMAX - l);
    }
    while (kraft > (1u << HUF_LMAX)) {
        /* lengthen: largest length < LMAX; ties -> smallest count, then largest symbol.
         * leaf[] is sorted by (count, symbol) ascending. */
        int best = -1;
        for (int i = 0; i < nz; i++) {
            const int l = lens[leaf[i].sym];
            if (l >= HUF_LMAX) continue;
            if (best < 0) { best = i; continue; }
            const int lb = lens[leaf[best].sym];
            if (l > lb) best = i;
            else if (l == lb) {
                if (leaf[i].cnt < leaf[best].cnt) best = i;
                else if (leaf[i].cnt == leaf[best].cnt && leaf[i].sym > leaf[best].sym) best = i;
            }
        }
        const int l = lens[leaf[best].sym];
        kraft -= 1u << (HUF_LMAX - l - 1);
        lens[leaf[best].sym] = (uint8_t)(l + 1);
    }
    for (;;) {
        /* shorten: most frequent symbol (ties -> smallest symbol) whose shortening still fits */
        int best = -1;
        for (int i = nz - 1; i >= 0; i--) {
            const int l = lens[leaf[i].sym];
            if (l <= 1) continue;
            if (kraft + (1u << (HUF_LMAX - l)) > (1u << HUF_LMAX)) continue;
            if (best < 0) { best = i; continue; }
            if (leaf[i].cnt > leaf[best].cnt) best = i;
            else if (leaf[i].cnt == leaf[best].cnt && leaf[i].sym < leaf[best].sym) best = i;
        }
        if (best < 0) break;
        const int l = lens[leaf[best].sym];
        kraft += 1u << (HUF_LMAX - l);
        lens[leaf[best].sym] = (uint8_t)(l - 1);
    }
}

/* canonical codes, bit-reversed to their length */
void huf_oracle_codes(const uint8_t lens[256], uint16_t codes[256])
{
    uint32_t count[HUF_LMAX + 2] = {0}, first[HUF_LMAX + 2] = {0};
    for (int s = 0; s < 256; s++) count[lens[s]]++;
    count[0] = 0;
    uint32_t code = 0;
    for (int l = 1; l <= HUF_LMAX; l++) { code = (code + count[l - 1]) << 1; first[l] = code; }
    for (int s = 0; s < 256; s++) {
        const int l = lens[s];
        codes[s] = 0;
        if (!l) continue;
        uint32_t c = first[l]++;
        uint32_t r = 0;
        for (int b = 0; b < l; b++) r |= ((c >> b) & 1u) << (l - 1 - b);
        codes[s] = (uint16_t)r;
    }
}

static void pack_table(const uint8_t lens[256], uint8_t table[128])
{
    for (int i = 0; i < 128; i++) table[i] = (uint8_t)(lens[2 * i] | (lens[2 * i + 1] << 4));
}
static void unpack_table(const uint8_t table[128], uint8_t lens[256])
{
    for (int i = 0; i < 128; i++) { lens[2 * i] = table[i] & 15; lens[2 * i + 1] = table[i] >> 4; }
}

/* ------------------------------------------------------------ container */

/* sizes[c] = exact bytes of chunk c's Sprintz stream at dense + offsets[c].
 * Writes tables (128 B per segment), records into out at out_offsets[c] (filled in here,
 * out_offsets[nchunks] = total).  Returns total bytes. */
uint64_t huf_oracle_compress(const uint8_t* dense, const uint64_t* offsets, const uint32_t* sizes, uint64_t nchunks,
                             uint8_t* out, uint64_t* out_offsets, uint8_t* tables)
{
    uint64_t pos = 0;
    for (uint64_t seg = 0; seg * HUF_SEG < nchunks; seg++) {
        const uint64_t c0 = seg * HUF_SEG, c1 = (c0 + HUF_SEG < nchunks) ? c0 + HUF_SEG : nchunks;
        uint32_t counts[256] = {0};
        for (uint64_t c = c0; c < c1; c++)
            for (uint32_t i = 0; i < sizes[c]; i++) counts[dense[offsets[c] + i]]++;
        uint8_t lens[256];
        uint16_t codes[256];
        huf_oracle_lengths(counts, lens);
        huf_oracle_codes(lens, codes);
        pack_table(lens, tables + seg * 128);
        for (uint64_t c = c0; c < c1; c++) {
            const uint8_t* s = dense + offsets[c];
            const uint32_t n = sizes[c];
            const uint32_t q = (n + 3) / 4;
            uint32_t sz[4];
            uint64_t enc = 0;
            for (int j = 0; j < 4; j++) {
                uint64_t bits = 0;
                const uint32_t a = (uint32_t)j * q < n ? (uint32_t)j * q : n, b = (a + q < n) ? a + q : n;
                for (uint32_t i = a; i < b; i++) bits += lens[s[i]];
                sz[j] = (uint32_t)((bits + 7) / 8);
                enc += sz[j];
            }
            pos = (pos + 3) & ~(uint64_t)3;
            out_offsets[c] = pos;
            uint8_t* o = out + pos;
            const int stored = (12 + enc >= 4 + (uint64_t)n) || sz[0] > 0xffff || sz[1] > 0xffff || sz[2] > 0xffff;
            const uint32_t hdr = n | (stored ? 0x80000000u : 0);
            memcpy(o, &hdr, 4);
            if (stored) {
                memcpy(o + 4, s, n);
                pos += 4 + (uint64_t)n;
                continue;
            }
            const uint16_t h[4] = {(uint16_t)sz[0], (uint16_t)sz[1], (uint16_t)sz[2], 0};
            memcpy(o + 4, h, 8);
            uint8_t* w = o + 12;
            for (int j = 0; j < 4; j++) {
                const uint32_t a = (uint32_t)j * q < n ? (uint32_t)j * q : n, b = (a + q < n) ? a + q : n;
                memset(w, 0, sz[j]);
                uint64_t bit = 0;
                for (uint32_t i = a; i < b; i++) {
                    const uint32_t code = codes[s[i]], l = lens[s[i]];
                    uint32_t x = code << (bit & 7);
                    uint64_t by = bit >> 3;
                    while (x) { w[by++] |= (uint8_t)x; x >>= 8; }
                    bit += l;
                }
                w += sz[j];
            }
            pos += 12 + enc;
        }
    }
    pos = (pos + 3) & ~(uint64_t)3;              /* the container ends on a 4-byte boundary too */
    out_offsets[nchunks] = pos;
    return pos;
}

/* Inverse: rebuilds the dense Sprintz container (chunk starts rounded up to `align`),
 * fills offsets[nchunks+1] and sizes[nchunks]. Returns total dense bytes. */
uint64_t huf_oracle_decompress(const uint8_t* huf, const uint64_t* huf_offsets, const uint8_t* tables, uint64_t nchunks,
                               uint32_t align, uint8_t* dense, uint64_t* offsets, uint32_t* sizes)
{
    uint64_t pos = 0;
    uint16_t* dtab = (uint16_t*)malloc(sizeof(uint16_t) << HUF_LMAX);   /* sym | len << 8 */
    for (uint64_t seg = 0; seg * HUF_SEG < nchunks; seg++) {
        const uint64_t c0 = seg * HUF_SEG, c1 = (c0 + HUF_SEG < nchunks) ? c0 + HUF_SEG : nchunks;
        uint8_t lens[256];
        uint16_t codes[256];
        unpack_table(tables + seg * 128, lens);
        huf_oracle_codes(lens, codes);
        memset(dtab, 0, sizeof(uint16_t) << HUF_LMAX);
        for (int s = 0; s < 256; s++) {
            const int l = lens[s];
            if (!l) continue;
            for (uint32_t k = codes[s]; k < (1u << HUF_LMAX); k += 1u << l) dtab[k] = (uint16_t)(s | (l << 8));
        }
        for (uint64_t c = c0; c < c1; c++) {
            const uint8_t* r = huf + huf_offsets[c];
            uint32_t hdr;
            memcpy(&hdr, r, 4);
            const uint32_t n = hdr & 0x7fffffffu;
            pos = (pos + align - 1) & ~(uint64_t)(align - 1);
            offsets[c] = pos;
            sizes[c] = n;
            uint8_t* o = dense + pos;
            if (hdr >> 31) { memcpy(o, r + 4, n); pos += n; continue; }
            uint16_t h[4];
            memcpy(h, r + 4, 8);
            const uint32_t q = (n + 3) / 4;
            const uint8_t* w = r + 12;
            const uint64_t rec_end = huf_offsets[c + 1];
            for (int j = 0; j < 4; j++) {
                const uint32_t a = (uint32_t)j * q < n ? (uint32_t)j * q : n, b = (a + q < n) ? a + q : n;
                const uint64_t avail = (j < 3) ? h[j] : (uint64_t)((huf + rec_end) - w);   /* bytes we may read */
                uint64_t bit = 0;
                for (uint32_t i = a; i < b; i++) {
                    uint32_t x = 0;
                    const uint64_t by = bit >> 3;
                    for (int k = 0; k < 3; k++) if (by + k < avail) x |= (uint32_t)w[by + k] << (8 * k);
                    const uint16_t e = dtab[(x >> (bit & 7)) & ((1u << HUF_LMAX) - 1)];
                    o[i] = (uint8_t)e;
                    bit += e >> 8;
                }
                if (j < 3) w += h[j];
            }
            pos += n;
        }
    }
    free(dtab);
    pos = (pos + align - 1) & ~(uint64_t)(align - 1);   /* like sprintz_mi355x_compact: the end is aligned too */
    offsets[nchunks] = pos;
    return pos;
}

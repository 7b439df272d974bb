// huf0.hip -- batched decoder for genuine Huff0 blocks (Yann Collet's Huff0 / FSE as shipped in
// zstd 1.4.x: HUF_compress's output = tree description + jump table + 4 bit streams), one block
// per chunk -- the entropy stage the paper applies after bit-packing (communicate/ubicomp/
// method.tex:293-297) in ITS wire format, where huf.hip is this repository's own GPU-shaped
// container.  The format is restated, with the library citations, in oracle/huf0_oracle.c;
// parity is pinned against the system libzstd's HUF_compress / HUF_decompress by the tests.
//
// A Huff0 block is 4-way parallel by construction and its code is private to the block: a wave
// takes 16 chunks, lane = (chunk, stream).  The library's decoder looks codes up in a
// 2^tableLog-entry table; 16 such tables are 64 KB of LDS and leave a CU two waves (measured:
// 4.6 ms on the headline shape, every phase latency-bound).  The code is canonical -- per code
// length ascending symbols, the longest codes lowest (HUF_readDTableX1's fill order) -- so the
// table is not needed: with start[w] = the first table index of weight w, a look-ahead value
// idx has weight w = 1 + #{k >= 2 : start[k] <= idx} (11 compare-and-adds on per-lane
// registers), its symbol is sorted[symoff[w] + ((idx - start[w]) >> (w - 1))] and it is
// tableLog + 1 - w bits long.  Per chunk that is 256 bytes of sorted symbols and 13 words in
// LDS instead of 4 KB, a dozen waves per CU instead of two, and every table log the format
// allows (12 included).  Headline shape (131 072 blocks of ~3.4 KB): 1.4 ms, of which phases
// A + B 0.5 and C 0.9 (its ~200 VALU per 4 symbols bound it at ~0.6).
// Phase A (first lane of each quad, serial): tree description -> weights; the FSE-compressed form
// is decoded with a 64-entry table in LDS.  Phase B (same lane): counting sort of the symbols by
// weight.  Phase C (every lane): its stream, read from the last byte down through a 64-bit
// window, 4 symbols per refill, refilled from 16-byte aligned pieces held in registers.
#include "../../include/sprintz_mi355x.h"

#include <hip/hip_runtime.h>

#include <string>

namespace sprintz { int set_error(int code, const char* what); }   // api.hip: the library's one error sink

namespace {

constexpr int kWStride = 256 + 4;                // weights per chunk, padded off the bank stride
constexpr int kRStride = 344 + 4;                // per chunk: 8 zero bytes + header copy 144 | norm 32 | next 32 | fse 128; later the sorted symbols (256)
constexpr int64_t kCorrupt = SPRINTZ_E_CORRUPT;

typedef uint32_t v2u __attribute__((ext_vector_type(2)));
typedef v2u __attribute__((aligned(1), may_alias)) v2u_a1;
typedef uint32_t __attribute__((aligned(1), may_alias)) u32_a1;

__device__ __forceinline__ int highbit(uint32_t v) { return 31 - __clz((int)v); }      // v != 0

// bytes b .. b+4 of a zero-padded LDS byte array, little endian
__device__ __forceinline__ uint64_t rd40(const uint8_t* h, uint32_t b)
{
    return (uint64_t)h[b] | ((uint64_t)h[b + 1] << 8) | ((uint64_t)h[b + 2] << 16) | ((uint64_t)h[b + 3] << 24) | ((uint64_t)h[b + 4] << 32);
}
// forward LSB-first reader (FSE_readNCount): 32 bits at bit position bp
__device__ __forceinline__ uint32_t fwd32(const uint8_t* h, uint32_t bp) { return (uint32_t)(rd40(h, bp >> 3) >> (bp & 7u)); }
// backward reader (BIT_DStream_t as a cursor P = unread bits): the nb (<= 16) bits below P, MSB first, 0 before the start
__device__ __forceinline__ uint32_t back_look(const uint8_t* h, int P, int nb)
{
    if (P <= 0 || nb == 0) return 0;
    const int lo = P - nb;
    if (lo >= 0) return (uint32_t)(rd40(h, (uint32_t)lo >> 3) >> (lo & 7)) & ((1u << nb) - 1u);
    return ((uint32_t)rd40(h, 0) & ((1u << P) - 1u)) << (-lo);
}

// HUF_readStats (entropy_common.c) over the header bytes h[0..n) (zero padded); weights[0..nsym).
// Scratch s: int16 norm[16] | u16 next[16] | u16 fse[64]: weights are < 12, so a description that
// gives probability to a symbol >= 16 is damaged.  Returns header bytes, 0 if damaged.
__device__ uint32_t read_stats(const uint8_t* h, uint32_t n, uint8_t* weights, uint8_t* s, uint32_t& nsym, uint32_t& tl_out)
{
    if (n < 1) return 0;
    uint32_t isize = h[0], osize;
    if (isize >= 128) {                                           // 4-bit weights
        osize = isize - 127;
        isize = (osize + 1) / 2;
        if (isize + 1 > n) return 0;
        for (uint32_t k = 0; k < osize; k += 2) {
            weights[k] = h[1 + k / 2] >> 4;
            weights[k + 1] = h[1 + k / 2] & 15;
        }
    } else {                                                      // FSE_decompress_wksp, table log <= 6
        if (isize + 1 > n) return 0;
        const uint8_t* const f = h + 1;
        int16_t* const norm = (int16_t*)s;
        uint16_t* const next = (uint16_t*)(s + 32);
        uint16_t* const fse = (uint16_t*)(s + 64);                // symbol | nbits << 4 | new_state << 8
        for (int k = 0; k < 16; k++) norm[k] = 0;
        // FSE_readNCount
        uint32_t bp = 0;
        int nb = (int)(fwd32(f, bp) & 0xf) + 5;
        if (nb > 6) return 0;                                     // tableLog > maxLog (6)
        bp += 4;
        const uint32_t tl = (uint32_t)nb;
        int remaining = (1 << nb) + 1, threshold = 1 << nb;
        nb++;
        uint32_t charnum = 0;
        bool previous0 = false;
        const uint32_t bit_end = 8u * isize;
        while (remaining > 1 && charnum <= 255u) {
            if (previous0) {
                uint32_t n0 = charnum;
                while ((fwd32(f, bp) & 0xffffu) == 0xffffu) { n0 += 24; bp += 16; if (bp > bit_end) return 0; }
                while ((fwd32(f, bp) & 3u) == 3u) { n0 += 3; bp += 2; if (bp > bit_end) return 0; }
                n0 += fwd32(f, bp) & 3u;
                bp += 2;
                if (n0 > 255u) return 0;
                charnum = n0;                                     // norm is zero there already
            }
            const uint32_t bits = fwd32(f, bp);
            const int max = (2 * threshold - 1) - remaining;
            int count;
            if ((int)(bits & (uint32_t)(threshold - 1)) < max) {
                count = (int)(bits & (uint32_t)(threshold - 1));
                bp += (uint32_t)(nb - 1);
            } else {
                count = (int)(bits & (uint32_t)(2 * threshold - 1));
                if (count >= threshold) count -= max;
                bp += (uint32_t)nb;
            }
            count--;
            remaining -= count < 0 ? -count : count;
            if (charnum > 255u || bp > bit_end) return 0;
            if (charnum >= 16u) { if (count != 0) return 0; charnum++; }
            else norm[charnum++] = (int16_t)count;
            previous0 = count == 0;
            while (remaining < threshold) { nb--; threshold >>= 1; }
        }
        if (remaining != 1 || bp > bit_end || charnum == 0) return 0;
        const uint32_t max_sv = (charnum < 16u ? charnum : 16u) - 1, hl = (bp + 7) >> 3;
        if (hl >= isize) return 0;
        // FSE_buildDTable
        const uint32_t size = 1u << tl;
        uint32_t high = size - 1;
        for (uint32_t sy = 0; sy <= max_sv; sy++) {
            if (norm[sy] == -1) { fse[high--] = (uint16_t)sy; next[sy] = 1; }
            else next[sy] = (uint16_t)norm[sy];
        }
        {
            const uint32_t mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
            uint32_t pos = 0, placed = 0;
            for (uint32_t sy = 0; sy <= max_sv; sy++)
                for (int i = 0; i < norm[sy]; i++) {
                    if (++placed > size) return 0;
                    fse[pos] = (uint16_t)sy;
                    pos = (pos + step) & mask;
                    while (pos > high) pos = (pos + step) & mask;
                }
            if (pos != 0) return 0;
        }
        for (uint32_t u = 0; u < size; u++) {
            const uint32_t sy = fse[u] & 0xfu, ns = next[sy]++;
            if (ns == 0 || ns >= 2 * size) return 0;
            const uint32_t nbits = tl - (uint32_t)highbit(ns);
            fse[u] = (uint16_t)(sy | (nbits << 4) | (((ns << nbits) - size) << 8));
        }
        // FSE_decompress_usingDTable: two interleaved states; the stream ends by running dry
        const uint8_t* const b = f + hl;
        const uint32_t bn = isize - hl;
        if (b[bn - 1] == 0) return 0;
        int P = 8 * (int)(bn - 1) + highbit(b[bn - 1]);
        uint32_t s1 = back_look(b, P, (int)tl); P -= (int)tl;
        uint32_t s2 = back_look(b, P, (int)tl); P -= (int)tl;
        osize = 0;
        // the stream's next 64 bits ride in a register, refilled every 8 weights (8 x 6 bits <= the 57 a refill guarantees)
        for (;;) {
            uint64_t win = 0;
            if (P > 0) {
                const int tb = (P - 1) >> 3;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int at = tb - 7 + k;
                    win |= (uint64_t)(at >= 0 ? b[at] : (uint8_t)0) << (8 * k);
                }
                win <<= 7 - ((P - 1) & 7);
                if (P < 64) win &= ~0ull << (64 - P);
            }
            bool done = false;
#pragma unroll
            for (int r = 0; r < 4 && !done; r++) {
                if (osize + 2 > 255u) return 0;
                uint32_t e = fse[s1];
                weights[osize++] = (uint8_t)(e & 0xfu);
                uint32_t nbt = (e >> 4) & 0xfu;
                s1 = (e >> 8) + (nbt ? (uint32_t)(win >> (64u - nbt)) : 0u);
                win <<= nbt;
                P -= (int)nbt;
                if (P < 0) { weights[osize++] = (uint8_t)(fse[s2] & 0xfu); done = true; break; }
                if (osize + 2 > 255u) return 0;
                e = fse[s2];
                weights[osize++] = (uint8_t)(e & 0xfu);
                nbt = (e >> 4) & 0xfu;
                s2 = (e >> 8) + (nbt ? (uint32_t)(win >> (64u - nbt)) : 0u);
                win <<= nbt;
                P -= (int)nbt;
                if (P < 0) { weights[osize++] = (uint8_t)(fse[s1] & 0xfu); done = true; break; }
            }
            if (done) break;
        }
    }
    // weight statistics; the last symbol's weight is implied
    uint32_t total = 0, rank1 = 0;
    for (uint32_t k = 0; k < osize; k++) {
        const uint32_t w = weights[k];
        if (w >= 12u) return 0;
        rank1 += w == 1u;
        total += (1u << w) >> 1;
    }
    if (total == 0) return 0;
    const uint32_t tl = (uint32_t)highbit(total) + 1u;
    if (tl > 12u) return 0;
    const uint32_t rest = (1u << tl) - total;
    if ((1u << highbit(rest)) != rest) return 0;
    const uint32_t lw = (uint32_t)highbit(rest) + 1u;
    weights[osize] = (uint8_t)lw;
    rank1 += lw == 1u;
    if (rank1 < 2 || (rank1 & 1u)) return 0;
    nsym = osize + 1;
    tl_out = tl;
    return isize + 1;
}

// ---- the same function as the tree kernel runs it: one lane per chunk, every lane of the wave busy, so what counts
// is the length of the DEPENDENT chain of LDS round trips.  Differences in form, none in result (every rejection of
// read_stats is kept): the header copy `hb` carries 8 zero bytes in front and zero padding behind, so bit fields are
// two aligned dwords + v_alignbit instead of five byte reads; the two FSE states' table reads are issued together;
// the weights are read back four at a time; the per-weight counts are LDS adds with no return (cnt[13], zeroed by
// the caller; on return cnt[w] = number of symbols of weight w, the implied last one included).
__device__ __forceinline__ uint32_t hb32(const uint8_t* hb, uint32_t bitpos)                       // 32 bits at bit `bitpos` of hb, LSB first
{
    const uint32_t* const q = (const uint32_t*)hb + (bitpos >> 5);
    return __builtin_amdgcn_alignbit(q[1], q[0], bitpos & 31u);
}
__device__ uint32_t read_stats2(const uint8_t* hb, uint32_t n, uint8_t* weights, uint8_t* s, uint32_t* cnt, uint32_t& nsym, uint32_t& tl_out)
{
    const uint8_t* const h = hb + 8;
    if (n < 1) return 0;
    uint32_t isize = h[0], osize;
    if (isize >= 128) {                                           // 4-bit weights
        osize = isize - 127;
        isize = (osize + 1) / 2;
        if (isize + 1 > n) return 0;
        for (uint32_t k = 0; k < osize; k += 2) {
            const uint32_t v = h[1 + k / 2];
            weights[k] = (uint8_t)(v >> 4);
            weights[k + 1] = (uint8_t)(v & 15u);
        }
    } else {                                                      // FSE_decompress_wksp, table log <= 6
        if (isize + 1 > n) return 0;
        int16_t* const norm = (int16_t*)s;
        uint16_t* const next = (uint16_t*)(s + 32);
        uint16_t* const fse = (uint16_t*)(s + 64);                // symbol | nbits << 4 | new_state << 8
        for (int k = 0; k < 8; k++) ((uint32_t*)norm)[k] = 0;
        // FSE_readNCount: bit 0 of the description is bit 72 of hb (8 lead bytes + the size byte)
        constexpr uint32_t F0 = 72;
        uint32_t bp = 0;
        int nb = (int)(hb32(hb, F0 + bp) & 0xf) + 5;
        if (nb > 6) return 0;                                     // tableLog > maxLog (6)
        bp += 4;
        const uint32_t tl = (uint32_t)nb;
        int remaining = (1 << nb) + 1, threshold = 1 << nb;
        nb++;
        uint32_t charnum = 0;
        bool previous0 = false;
        const uint32_t bit_end = 8u * isize;
        while (remaining > 1 && charnum <= 255u) {
            if (previous0) {
                uint32_t n0 = charnum;
                while ((hb32(hb, F0 + bp) & 0xffffu) == 0xffffu) { n0 += 24; bp += 16; if (bp > bit_end) return 0; }
                while ((hb32(hb, F0 + bp) & 3u) == 3u) { n0 += 3; bp += 2; if (bp > bit_end) return 0; }
                n0 += hb32(hb, F0 + bp) & 3u;
                bp += 2;
                if (n0 > 255u) return 0;
                charnum = n0;                                     // norm is zero there already
            }
            const uint32_t bits = hb32(hb, F0 + bp);
            const int max = (2 * threshold - 1) - remaining;
            int count;
            if ((int)(bits & (uint32_t)(threshold - 1)) < max) {
                count = (int)(bits & (uint32_t)(threshold - 1));
                bp += (uint32_t)(nb - 1);
            } else {
                count = (int)(bits & (uint32_t)(2 * threshold - 1));
                if (count >= threshold) count -= max;
                bp += (uint32_t)nb;
            }
            count--;
            remaining -= count < 0 ? -count : count;
            if (charnum > 255u || bp > bit_end) return 0;
            if (charnum >= 16u) { if (count != 0) return 0; charnum++; }
            else norm[charnum++] = (int16_t)count;
            previous0 = count == 0;
            while (remaining < threshold) { nb--; threshold >>= 1; }
        }
        if (remaining != 1 || bp > bit_end || charnum == 0) return 0;
        const uint32_t max_sv = (charnum < 16u ? charnum : 16u) - 1, hl = (bp + 7) >> 3;
        if (hl >= isize) return 0;
        // FSE_buildDTable
        const uint32_t size = 1u << tl;
        uint32_t high = size - 1;
        for (uint32_t sy = 0; sy <= max_sv; sy++) {
            if (norm[sy] == -1) { fse[high--] = (uint16_t)sy; next[sy] = 1; }
            else next[sy] = (uint16_t)norm[sy];
        }
        {
            const uint32_t mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
            uint32_t pos = 0, placed = 0;
            for (uint32_t sy = 0; sy <= max_sv; sy++)
                for (int i = 0; i < norm[sy]; i++) {
                    if (++placed > size) return 0;
                    fse[pos] = (uint16_t)sy;
                    pos = (pos + step) & mask;
                    while (pos > high) pos = (pos + step) & mask;
                }
            if (pos != 0) return 0;
        }
        for (uint32_t u = 0; u < size; u++) {
            const uint32_t sy = fse[u] & 0xfu, ns = next[sy]++;
            if (ns == 0 || ns >= 2 * size) return 0;
            const uint32_t nbits = tl - (uint32_t)highbit(ns);
            fse[u] = (uint16_t)(sy | (nbits << 4) | (((ns << nbits) - size) << 8));
        }
        // FSE_decompress_usingDTable: two interleaved states; the stream ends by running dry
        const uint8_t* const b = h + 1 + hl;
        const uint32_t bn = isize - hl;
        if (b[bn - 1] == 0) return 0;
        int P = 8 * (int)(bn - 1) + highbit(b[bn - 1]);
        uint32_t s1 = back_look(b, P, (int)tl); P -= (int)tl;
        uint32_t s2 = back_look(b, P, (int)tl); P -= (int)tl;
        osize = 0;
        const uint32_t B0 = 8u + 1u + hl;                         // byte offset of the bit stream in hb
        // the stream's next 64 bits ride in a register, refilled every 8 weights (8 x 6 bits <= the 57 a refill guarantees)
        bool done = false;
        while (!done) {
            uint64_t win = 0;
            if (P > 0) {
                const uint32_t a = B0 + ((uint32_t)(P - 1) >> 3) - 7u;            // first of the 8 bytes that end at the cursor's byte (>= 2)
                const uint32_t* const q = (const uint32_t*)hb + (a >> 2);
                const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
                const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, a & 3u), hi = __builtin_amdgcn_alignbyte(d2, d1, a & 3u);
                win = (((uint64_t)hi << 32) | lo) << (7 - ((P - 1) & 7));
                if (P < 64) win &= ~0ull << (64 - P);             // nothing before the stream's first bit
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t e1 = fse[s1], e2 = fse[s2];       // both table reads in flight together
                if (osize + 2 > 255u) return 0;
                weights[osize++] = (uint8_t)(e1 & 0xfu);
                uint32_t nbt = (e1 >> 4) & 0xfu;
                s1 = (e1 >> 8) + ((((uint32_t)(win >> 32)) >> 1) >> (31u - nbt));
                win <<= nbt;
                P -= (int)nbt;
                if (P < 0) { weights[osize++] = (uint8_t)(e2 & 0xfu); done = true; break; }
                if (osize + 2 > 255u) return 0;
                weights[osize++] = (uint8_t)(e2 & 0xfu);
                nbt = (e2 >> 4) & 0xfu;
                s2 = (e2 >> 8) + ((((uint32_t)(win >> 32)) >> 1) >> (31u - nbt));
                win <<= nbt;
                P -= (int)nbt;
                if (P < 0) { weights[osize++] = (uint8_t)(fse[s1] & 0xfu); done = true; break; }
            }
        }
    }
    // weight statistics, four weights a read; the last symbol's weight is implied
    uint32_t total = 0, rank1 = 0;
    bool heavy = false;
    for (uint32_t k = 0; k < osize; k += 4) {
        const uint32_t four = *(const uint32_t*)(weights + k);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t w = (four >> (8 * i)) & 0xffu;
            if (k + i < osize) {
                heavy |= w >= 12u;
                rank1 += w == 1u;
                total += (1u << (w & 31u)) >> 1;
                __hip_atomic_fetch_add(&cnt[w < 12u ? w : 0u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        }
    }
    if (heavy) return 0;
    if (total == 0) return 0;
    const uint32_t tl = (uint32_t)highbit(total) + 1u;
    if (tl > 12u) return 0;
    const uint32_t rest = (1u << tl) - total;
    if ((1u << highbit(rest)) != rest) return 0;
    const uint32_t lw = (uint32_t)highbit(rest) + 1u;
    weights[osize] = (uint8_t)lw;
    __hip_atomic_fetch_add(&cnt[lw], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    rank1 += lw == 1u;
    if (rank1 < 2 || (rank1 & 1u)) return 0;
    nsym = osize + 1;
    tl_out = tl;
    return isize + 1;
}

__device__ __forceinline__ int quad_bcast0(int v) { return __builtin_amdgcn_mov_dpp(v, 0x00, 0xf, 0xf, true); }
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------------------------
// Two launches.  (1) huf0_tree_kernel: ONE LANE PER CHUNK (64 chunks a wave) turns every coded
// block's tree description into a 320-byte descriptor in global memory: the symbols sorted by
// (weight, symbol), start[w] | symoff[w] << 16 for w = 1..12, and the header length / table log.
// Phases A and B used to run on the first lane of each quad of the stream decoder -- a quarter of
// the lanes for a third of its instructions (0.5 of 1.4 ms on the headline shape).
// (2) huf0_stream_kernel: lane = (chunk, stream), 16 chunks a wave, as before -- but a symbol is no
// longer an 11-step weight search.  The top 8 bits of the look-ahead index a per-chunk 256-entry
// table in LDS that resolves every code of <= 8 bits (symbol | length << 8); longer codes can only
// have the weights 1 .. tableLog - 8 <= 4, so THEIR weight is three compares against start[2..4],
// and their symbol one more LDS read from the sorted list.  Both reads are issued together; the
// dependent chain of a symbol is one LDS round trip.
constexpr int kDescStride = 320;                 // sorted[256] | u32 tab[16]: [0] = hl | tl << 16, [w] = start[w] | symoff[w] << 16
constexpr int kCStride = 256 + 64 + 512 + 4;     // stream kernel, per chunk in LDS: sorted | tab | table8; odd in dwords
constexpr int kRingStride = 32 + 8;              // stream kernel, per lane: two 16-byte pieces of its stream (8-byte aligned)

__global__ void __launch_bounds__(64) huf0_tree_kernel(const uint8_t* __restrict__ blocks, const uint64_t* __restrict__ boffs,
                                                       const uint64_t* __restrict__ ooffs, uint64_t nchunks, uint8_t* __restrict__ desc)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_w[64 * kWStride];
    __shared__ __attribute__((aligned(16))) uint8_t s_r[64 * kRStride];
    __shared__ uint32_t s_tab[64][17];
    __shared__ uint32_t s_run[64][13];
    __shared__ uint64_t s_src[64];
    __shared__ uint32_t s_hc[64];
    const int t = threadIdx.x;
    const uint64_t chunk0 = (uint64_t)blockIdx.x * 64;
    const uint64_t chunk = chunk0 + (uint64_t)t;
    const bool exists = chunk < nchunks;
    const uint64_t b0 = exists ? boffs[chunk] : 0, b1 = exists ? boffs[chunk + 1] : 0;
    const uint64_t o0 = exists ? ooffs[chunk] : 0, o1 = exists ? ooffs[chunk + 1] : 0;
    const uint64_t csize = b1 - b0, dsize = o1 - o0;
    const bool coded = exists && b1 >= b0 && o1 >= o0 && csize > 1 && csize < dsize;      // HUF_decompress's third case
    const uint32_t hcopy = coded ? (uint32_t)(csize < 129 ? csize : 129) : 0u;
    s_src[t] = (uint64_t)(uintptr_t)(blocks + b0);
    s_hc[t] = hcopy;
    wave_sync();
    // the wave copies one chunk's header per trip: 38 lanes, one (unaligned) dword each -- 8 zero bytes, the 129 header bytes
    // (one request), zero padding up to byte 152
    for (int c0 = 0; c0 < 64; c0 += 8) {                          // eight chunks a trip, their loads in flight together
        uint32_t v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t hc = s_hc[c0 + i];
            const uint8_t* const src = (const uint8_t*)(uintptr_t)s_src[c0 + i];
            v[i] = 0;
            if (t >= 2 && t < 38) {
                const uint32_t k = 4u * (uint32_t)(t - 2);
                if (k + 4 <= hc) v[i] = *(const u32_a1*)(src + k);
                else for (uint32_t bb = 0; bb < 4 && k + bb < hc; bb++) v[i] |= (uint32_t)src[k + bb] << (8 * bb);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; i++)
            if (t < 38) *(uint32_t*)(s_r + (c0 + i) * kRStride + 4u * (uint32_t)t) = v[i];
    }
#pragma unroll
    for (int w = 0; w < 13; w++) s_tab[t][w] = 0;                 // the per-weight counts
    wave_sync();
    uint8_t* const wts = s_w + t * kWStride;
    uint8_t* const scratch = s_r + t * kRStride;
    uint8_t* const sorted = scratch;
    uint32_t hl = 0, nsym = 0, tl = 0;
    if (coded) {
        hl = read_stats2(scratch, hcopy, wts, scratch + 152, &s_tab[t][0], nsym, tl);
        if (hl >= csize) hl = 0;
    }
    if (hl) {
        // start[w] (first table index of weight w), symoff[w], and the symbols sorted by (weight, symbol)
        uint32_t cnt[13];
#pragma unroll
        for (int w = 1; w < 13; w++) cnt[w] = s_tab[t][w];
        uint32_t at = 0, so = 0;
#pragma unroll
        for (int w = 1; w < 13; w++) {
            s_tab[t][w] = at | (so << 16);
            s_run[t][w] = so;
            at += cnt[w] << (w - 1);
            so += cnt[w];
        }
        // counting sort, four symbols a trip: their slots come back from four LDS adds issued together
        // (DS operations of a wave execute in issue order, so equal weights keep their symbol order)
        for (uint32_t sy = 0; sy < nsym; sy += 4) {
            const uint32_t four = *(const uint32_t*)(wts + sy);
            uint32_t pos[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t w = (four >> (8 * i)) & 0xffu;
                pos[i] = 0xffffffffu;
                if (sy + i < nsym && w) pos[i] = __hip_atomic_fetch_add(&s_run[t][w], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (pos[i] != 0xffffffffu) sorted[pos[i] & 0xffu] = (uint8_t)(sy + i);
        }
    }
    s_tab[t][0] = hl | (tl << 16);
    wave_sync();
    // descriptors out: 64 dwords of sorted symbols + 13 words of table per chunk, one chunk per trip
    for (int c = 0; c < 64; c++) {
        if (chunk0 + (uint64_t)c >= nchunks) break;
        uint8_t* const d = desc + (chunk0 + (uint64_t)c) * kDescStride;
        const uint32_t v = *(const uint32_t*)(s_r + c * kRStride + 4 * t);
        *(uint32_t*)(d + 4 * t) = v;
        if (t < 13) *(uint32_t*)(d + 256 + 4 * t) = s_tab[c][t];
    }
}

__global__ void __launch_bounds__(64) huf0_stream_kernel(const uint8_t* __restrict__ blocks, const uint64_t* __restrict__ boffs,
                                                         uint64_t nchunks, uint8_t* __restrict__ out,
                                                         const uint64_t* __restrict__ ooffs, int64_t* __restrict__ rets,
                                                         const uint8_t* __restrict__ desc)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_c[16 * kCStride];
    __shared__ __attribute__((aligned(16))) uint8_t s_ring[64 * kRingStride];
    const int t = threadIdx.x, q = t >> 2, j = t & 3;
    const uint64_t chunk0 = (uint64_t)blockIdx.x * 16;
    const uint64_t chunk = chunk0 + (uint64_t)q;
    const bool exists = chunk < nchunks;
    const uint64_t b0 = exists ? boffs[chunk] : 0, b1 = exists ? boffs[chunk + 1] : 0;
    const uint64_t o0 = exists ? ooffs[chunk] : 0, o1 = exists ? ooffs[chunk + 1] : 0;
    const uint8_t* const src = blocks + b0;
    uint8_t* const dst = out + o0;
    const uint64_t csize = b1 - b0, dsize = o1 - o0;
    uint8_t* const cbase = s_c + q * kCStride;
    const uint8_t* const sorted = cbase;
    const uint32_t* const tab = (const uint32_t*)(cbase + 256);
    uint16_t* const table8 = (uint16_t*)(cbase + 320);

    // ---- descriptors of this wave's 16 chunks -> LDS: 320 pieces of 16 bytes, five per lane
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    typedef v4u __attribute__((aligned(1), may_alias)) v4u_a1;
    typedef v4u __attribute__((aligned(16), may_alias)) v4u_a16;
#pragma unroll
    for (int r = 0; r < 5; r++) {
        const uint32_t piece = (uint32_t)r * 64 + (uint32_t)t, c = piece / 20u, part = piece % 20u;
        v4u v = {0, 0, 0, 0};
        if (chunk0 + c < nchunks) v = *(const v4u_a16*)(desc + (chunk0 + c) * kDescStride + 16u * part);
        uint32_t* const d = (uint32_t*)(s_c + c * kCStride + 16u * part);
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }

    // ---- HUF_decompress's conventions (huf_decompress.c): stored, one repeated byte, or a coded block
    int mode = 0;                                                 // 0 nothing / damaged, 1 stored, 2 repeated byte, 3 coded
    int64_t ret = 0;
    if (exists) {
        if (dsize == 0) ret = csize == 0 ? 0 : kCorrupt;
        else if (b1 < b0 || o1 < o0 || csize == 0 || csize > dsize) ret = kCorrupt;
        else if (csize == dsize) mode = 1;
        else if (csize == 1) mode = 2;
        else mode = 3;
    }
    if (mode == 1) for (uint64_t k = (uint64_t)j; k < dsize; k += 4) dst[k] = src[k];
    if (mode == 2) { const uint8_t v = src[0]; for (uint64_t k = (uint64_t)j; k < dsize; k += 4) dst[k] = v; }
    wave_sync();
    const uint32_t hl = mode == 3 ? (tab[0] & 0xffffu) : 0u, tl = mode == 3 ? (tab[0] >> 16) : 0u;
    if (mode == 3 && hl == 0) ret = kCorrupt;
    const bool coded = mode == 3 && hl != 0;

    // ---- table8: entry b = the code that starts with the 8 bits b, if it is at most 8 bits long.  Lane j of the
    // quad fills entries 64 j .. 64 j + 63; the weight only ever grows along them.
    if (coded) {
        const uint32_t start_short = tl > 8u ? (tab[tl - 7u] & 0xffffu) : 0u;      // first index of the codes of <= 8 bits
        uint32_t w = 1, e = tab[1], nxt = tab[2] & 0xffffu;
        for (uint32_t k = 0; k < 64; k++) {
            const uint32_t b = 64u * (uint32_t)j + k;
            const uint32_t idx0 = tl >= 8u ? b << (tl - 8u) : b >> (8u - tl);
            while (w < 12u && idx0 >= nxt) { w++; e = tab[w]; nxt = w < 12u ? (tab[w + 1] & 0xffffu) : 0xffffffffu; }
            const uint32_t pos = (e >> 16) + ((idx0 - (e & 0xffffu)) >> (w - 1u));
            uint32_t entry;
            if (idx0 >= start_short) entry = sorted[pos & 0xffu] | ((tl + 1u - w) << 8);                    // the code is <= 8 bits: symbol | length << 8
            else if (idx0 + (1u << (tl - 8u)) <= nxt) entry = 0x8000u | ((w - 1u) << 8) | (pos & 0xffu);   // longer codes, all of weight w: where they start in `sorted`
            else entry = 0xffffu;                                                                         // codes of several weights share these 8 bits: the general search
            table8[b] = (uint16_t)entry;
        }
    }
    wave_sync();

    // ---- lane j decodes stream j (HUF_decompress4X1_usingDTable_internal).  Set-up per lane, then ONE
    // wave-uniform loop: the quad exchanges of the output path need every lane, streamless ones included.
    bool bad = false;
    // long codes (> 8 bits) have weight <= tableLog - 8 <= 4: their weight is 1 + #{k in 2..4 : start[k] <= idx}
    const uint32_t E2 = coded ? tab[2] : 0xffffu, E3 = coded ? tab[3] : 0xffffu, E4 = coded ? tab[4] : 0xffffu;
    const uint32_t T2 = E2 & 0xffffu, T3 = E3 & 0xffffu, T4 = E4 & 0xffffu;
    const uint8_t* sp = blocks;                                   // this lane's stream
    int32_t P = 0;
    uint8_t* op = dst;
    uint64_t left = 0;
    if (coded) {
        const uint8_t* const ip = src + hl;
        const uint64_t n = csize - hl;
        if (n < 10) bad = true;
        uint64_t l[4] = {0, 0, 0, 0};
        if (!bad) {
            l[0] = (uint64_t)ip[0] | ((uint64_t)ip[1] << 8);
            l[1] = (uint64_t)ip[2] | ((uint64_t)ip[3] << 8);
            l[2] = (uint64_t)ip[4] | ((uint64_t)ip[5] << 8);
            if (6 + l[0] + l[1] + l[2] > n) bad = true;
            else l[3] = n - 6 - l[0] - l[1] - l[2];
            if (l[3] >= (1ull << 27)) bad = true;                 // the bit cursor is 32 bits: streams below 128 MiB (a Huff0 block is at most 128 KB)
        }
        if (!bad) {
            const uint64_t seg = (dsize + 3) / 4;
            uint64_t so = 6;
#pragma unroll
            for (int k = 0; k < 3; k++) so += k < j ? l[k] : 0;
            const uint64_t slen = j == 0 ? l[0] : j == 1 ? l[1] : j == 2 ? l[2] : l[3];
            uint64_t w0 = seg * (uint64_t)j;
            w0 = w0 < dsize ? w0 : dsize;
            const uint64_t w1 = j == 3 ? dsize : (w0 + seg < dsize ? w0 + seg : dsize);
            if (slen < 1 || ip[so + slen - 1] == 0) bad = true;
            if (!bad) {
                sp = ip + so;
                P = 8 * (int32_t)(slen - 1) + highbit(sp[slen - 1]);
                op = dst + w0;
                left = w1 - w0;
            }
        }
    }
    const bool streaming = left > 0 || (coded && !bad);           // has a stream whose end must be checked
    const uint32_t look_shift = 32u - (tl ? tl : 1u);
    const uint32_t bmask = tl > 8u ? (1u << (tl - 8u)) - 1u : 0u;     // the index bits below the 8-bit prefix
    const uint32_t t8 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)table8;
    const uint32_t so8 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)sorted;
    // The stream is read from its last byte down.  It reaches the lane as 16-byte aligned PIECES parked in a per-lane
    // LDS ring of two pieces (the 8-byte window never spans more); the piece below the ring is in flight in a register
    // and is only touched -- written into the ring slot the cursor just left -- when the cursor crosses into the next
    // piece, ~3 steps after it was requested.  (The first version rotated three pieces through registers: the
    // compiler's copies for the rotation made every step wait for the piece requested one step earlier.  A ring of
    // two 64-byte blocks hid the latency better but cost 8.7 KB of LDS a wave: 7 waves a CU instead of 10.)
    // Byte i of the stream sits at ring offset (s_al + i) & 31; piece k covers stream bytes [16 k - s_al, 16 k - s_al + 16).
    const uint32_t s_al = (uint32_t)((uintptr_t)sp & 15u);
    const uint8_t* const sp_al = sp - s_al;                       // >= blocks: the API asks for a 16-byte aligned buffer
    const int32_t last_piece = streaming ? (int32_t)((((P > 0 ? (uint32_t)(P - 1) >> 3 : 0u)) + s_al) >> 4) : -1;
    auto load_piece = [&](int32_t k) -> v4u {                     // pieces outside the stream read as zero, never touched
        if (k < 0 || k > last_piece) return v4u{0, 0, 0, 0};
        return *(const v4u_a16*)(sp_al + 16 * (int64_t)k);
    };
    const uint32_t ring = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)s_ring + (uint32_t)t * kRingStride;
    auto park = [&](int32_t pk, const v4u& f) {                   // piece pk -> its ring slot
        const uint32_t at = ring + (((uint32_t)pk & 1u) << 4);
        typedef __attribute__((address_space(3))) uint64_t lds_u64;
        *(lds_u64*)(uintptr_t)at = (uint64_t)f.x | ((uint64_t)f.y << 32);
        *(lds_u64*)(uintptr_t)(at + 8u) = (uint64_t)f.z | ((uint64_t)f.w << 32);
    };
    int32_t cur_b = last_piece;                                   // piece of the cursor's byte
    v4u fl;
    {
        const v4u f0 = load_piece(cur_b), f1 = load_piece(cur_b - 1);
        fl = load_piece(cur_b - 2);
        park(cur_b, f0);
        park(cur_b - 1, f1);
    }
    wave_sync();
    typedef __attribute__((address_space(3))) const uint16_t lds_u16;
    typedef __attribute__((address_space(3))) const uint8_t lds_u8c;
    typedef __attribute__((address_space(3))) const uint32_t lds_u32c;
    // one step = up to 4 symbols = one dword of output; branch-free except for the block crossing
    auto step = [&](uint32_t m) -> uint32_t {
        const int32_t Pc = P;
        const uint32_t x = ((uint32_t)(Pc > 0 ? Pc - 1 : 0) >> 3) + s_al;      // the cursor's byte, counted from piece 0
        if ((int32_t)(x >> 4) < cur_b) {                          // crossed into the piece below: the one in flight takes the freed slot
            park(cur_b - 2, fl);
            cur_b--;
            fl = load_piece(cur_b - 2);
        }
        // the 8 bytes ending at byte x: three aligned dwords of the ring, two v_alignbyte
        const uint32_t o = x - 7u;                                // may be "negative": bytes before the stream read as what the ring holds, masked below
        const uint32_t d0 = *(lds_u32c*)(uintptr_t)(ring + (o & 28u));
        const uint32_t d1 = *(lds_u32c*)(uintptr_t)(ring + ((o + 4u) & 28u));
        const uint32_t d2 = *(lds_u32c*)(uintptr_t)(ring + ((o + 8u) & 28u));
        const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, o & 3u), hi0 = __builtin_amdgcn_alignbyte(d2, d1, o & 3u);
        uint64_t win = (((uint64_t)hi0 << 32) | lo) << (7 - (int)((uint32_t)(Pc - 1) & 7u));
        if (__ballot(Pc < 64) != 0) {                             // only the last steps of a stream: nothing before its first bit
            const uint64_t keep = Pc >= 64 ? ~0ull : (Pc > 0 ? ~0ull << (64 - Pc) : 0ull);
            win &= keep;
        }
        uint32_t word = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t hi = (uint32_t)(win >> 32);
            const uint32_t e8 = *(lds_u16*)(uintptr_t)(t8 + ((hi >> 24) << 1));
            const uint32_t idx = hi >> look_shift;
            // a longer code whose 8-bit prefix holds one weight only: its symbol sits at base + (low index bits >> (w - 1))
            uint32_t wm1 = (e8 >> 8) & 3u;
            uint32_t pos = (e8 & 0xffu) + ((idx & bmask) >> wm1);
            if (__ballot(e8 == 0xffffu) != 0) {                   // rare: prefixes shared by several weights -- weight = 1 + #{k in 2..4 : start[k] <= idx}
                const bool g2 = idx >= T2, g3 = idx >= T3, g4 = idx >= T4;
                const uint32_t el = g4 ? E4 : (g3 ? E3 : (g2 ? E2 : 0u));
                const uint32_t wg = (uint32_t)g2 + (uint32_t)g3 + (uint32_t)g4;
                const bool mixed = e8 == 0xffffu;
                pos = mixed ? (el >> 16) + ((idx - (el & 0xffffu)) >> wg) : pos;
                wm1 = mixed ? wg : wm1;
            }
            const uint32_t syl = *(lds_u8c*)(uintptr_t)(so8 + (pos & 0xffu));
            const bool is_short = (e8 & 0x8000u) == 0;
            const bool on = (uint32_t)k < m;
            const uint32_t sym = is_short ? (e8 & 0xffu) : syl;
            const uint32_t nb = on ? (is_short ? (e8 >> 8) : (tl - wm1)) : 0u;
            word |= (on ? sym : 0u) << (8 * k);
            win <<= nb;
            P -= (int32_t)nb;
        }
        return word;
    };
    // The output leaves 64 bytes at a time: a lane collects 16 steps in registers, the quad transposes
    // its 16-byte pieces (two DPP butterfly stages) and every store writes ONE stream's 64 contiguous
    // bytes (4-byte stores per lane per step -- 524 288 open lines at the headline shape -- made this
    // phase 3.2 ms of a 3.7 ms launch).  A stream's final partial burst goes out narrow.
    const bool odd1 = (t & 1) != 0, odd2 = (t & 2) != 0;
    const uint32_t part = (uint32_t)t & 3u;
    for (;;) {
        if (__ballot(left > 0) == 0) break;
        const bool full = left >= 64;
        uint32_t wb[16];
#pragma unroll
        for (int sN = 0; sN < 16; sN++) {
            const uint32_t done = 4u * sN;
            const uint32_t m = (left > done && P >= -64) ? (left - done < 4 ? (uint32_t)(left - done) : 4u) : 0u;
            wb[sN] = step(m);
        }
        const uint32_t cnt = left < 64 ? (uint32_t)left : 64u;
        uint32_t v[4][4];
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int d = 0; d < 4; d++) v[k][d] = wb[4 * k + d];
#pragma unroll
        for (int k = 0; k < 4; k += 2)
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const uint32_t send = odd1 ? v[k][d] : v[k + 1][d];
                const uint32_t recv = (uint32_t)__builtin_amdgcn_mov_dpp((int)send, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
                if (odd1) v[k][d] = recv; else v[k + 1][d] = recv;
            }
#pragma unroll
        for (int k = 0; k < 2; k++)
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const uint32_t send = odd2 ? v[k][d] : v[k + 2][d];
                const uint32_t recv = (uint32_t)__builtin_amdgcn_mov_dpp((int)send, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
                if (odd2) v[k][d] = recv; else v[k + 2][d] = recv;
            }
        const uint64_t mine = full ? (uint64_t)(uintptr_t)op : 0ull;
        const int mlo = (int)(uint32_t)mine, mhi = (int)(uint32_t)(mine >> 32);
#pragma unroll
        for (int qq = 0; qq < 4; qq++) {
            uint32_t dlo, dhi;
            if (qq == 0) { dlo = (uint32_t)__builtin_amdgcn_mov_dpp(mlo, 0x00, 0xf, 0xf, true); dhi = (uint32_t)__builtin_amdgcn_mov_dpp(mhi, 0x00, 0xf, 0xf, true); }
            else if (qq == 1) { dlo = (uint32_t)__builtin_amdgcn_mov_dpp(mlo, 0x55, 0xf, 0xf, true); dhi = (uint32_t)__builtin_amdgcn_mov_dpp(mhi, 0x55, 0xf, 0xf, true); }
            else if (qq == 2) { dlo = (uint32_t)__builtin_amdgcn_mov_dpp(mlo, 0xAA, 0xf, 0xf, true); dhi = (uint32_t)__builtin_amdgcn_mov_dpp(mhi, 0xAA, 0xf, 0xf, true); }
            else { dlo = (uint32_t)__builtin_amdgcn_mov_dpp(mlo, 0xFF, 0xf, 0xf, true); dhi = (uint32_t)__builtin_amdgcn_mov_dpp(mhi, 0xFF, 0xf, 0xf, true); }
            const uint64_t da = ((uint64_t)dhi << 32) | dlo;
            if (da) {
                v4u piece = {v[qq][0], v[qq][1], v[qq][2], v[qq][3]};
                __builtin_nontemporal_store(piece, (v4u_a1*)(uintptr_t)(da + 16u * part));   // streamed out once
            }
        }
        if (!full && cnt) {
#pragma unroll
            for (int sN = 0; sN < 16; sN++) {
                const uint32_t done = 4u * sN;
                if (cnt >= done + 4) *(u32_a1*)(op + done) = wb[sN];
                else if (cnt > done) for (uint32_t k = 0; k < cnt - done; k++) op[done + k] = (uint8_t)(wb[sN] >> (8 * k));
            }
        }
        op += cnt;
        left -= cnt;
    }
    if (streaming && P != 0) bad = true;                          // every stream ends exactly (BIT_endOfDStream)
    const bool any_bad = __builtin_amdgcn_mov_dpp((int)bad, 0x00, 0xf, 0xf, true) | __builtin_amdgcn_mov_dpp((int)bad, 0x55, 0xf, 0xf, true) |
                         __builtin_amdgcn_mov_dpp((int)bad, 0xAA, 0xf, 0xf, true) | __builtin_amdgcn_mov_dpp((int)bad, 0xFF, 0xf, 0xf, true);
    if (exists && j == 0 && rets) {
        if (mode == 1 || mode == 2) ret = (int64_t)dsize;
        else if (mode == 3 && ret == 0) ret = any_bad ? kCorrupt : (int64_t)dsize;
        rets[chunk] = ret;
    }
}

std::string g_err0;

}  // namespace

extern "C" {

size_t sprintz_mi355x_huf0_decode_tmp_bytes(uint64_t nchunks) { return (size_t)nchunks * kDescStride + 256; }

int sprintz_mi355x_huf0_decompress_batch_ws(const void* d_blocks, const uint64_t* d_block_offsets, uint64_t nchunks, void* d_out,
                                            const uint64_t* d_out_offsets, int64_t* d_rets, void* d_tmp, void* hip_stream)
{
    if (!d_blocks || !d_block_offsets || !d_out || !d_out_offsets || ((uintptr_t)d_blocks & 15) || (nchunks && (!d_tmp || ((uintptr_t)d_tmp & 15))))
        return sprintz::set_error(SPRINTZ_E_INVALID, "Huff0 stage: invalid argument (null pointer, alignment or size)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return sprintz::set_error(SPRINTZ_E_NO_DEVICE, "Huff0 stage: no usable HIP device (there is no CPU fallback)");
    if (nchunks == 0) return 0;
    const uint64_t grid1 = (nchunks + 63) / 64, grid2 = (nchunks + 15) / 16;
    if (grid2 > 0x7fffffffull) return sprintz::set_error(SPRINTZ_E_INVALID, "Huff0 stage: invalid argument (null pointer, alignment or size)");
    hipStream_t st = (hipStream_t)hip_stream;
    hipLaunchKernelGGL(huf0_tree_kernel, dim3((unsigned)grid1), dim3(64), 0, st, (const uint8_t*)d_blocks, d_block_offsets, d_out_offsets,
                       nchunks, (uint8_t*)d_tmp);
    hipLaunchKernelGGL(huf0_stream_kernel, dim3((unsigned)grid2), dim3(64), 0, st, (const uint8_t*)d_blocks, d_block_offsets, nchunks,
                       (uint8_t*)d_out, d_out_offsets, d_rets, (const uint8_t*)d_tmp);
    return hipGetLastError() == hipSuccess ? 0 : sprintz::set_error(SPRINTZ_E_HIP, "Huff0 stage: a HIP call or kernel launch failed");
}

// the ABI-1 form without a workspace argument: stream-ordered allocation of the descriptors
int sprintz_mi355x_huf0_decompress_batch(const void* d_blocks, const uint64_t* d_block_offsets, uint64_t nchunks, void* d_out,
                                         const uint64_t* d_out_offsets, int64_t* d_rets, void* hip_stream)
{
    if (!d_blocks || !d_block_offsets || !d_out || !d_out_offsets || ((uintptr_t)d_blocks & 15))
        return sprintz::set_error(SPRINTZ_E_INVALID, "Huff0 stage: invalid argument (null pointer, alignment or size)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return sprintz::set_error(SPRINTZ_E_NO_DEVICE, "Huff0 stage: no usable HIP device (there is no CPU fallback)");
    if (nchunks == 0) return 0;
    void* tmp = nullptr;
    if (hipMallocAsync(&tmp, sprintz_mi355x_huf0_decode_tmp_bytes(nchunks), (hipStream_t)hip_stream) != hipSuccess)
        return sprintz::set_error(SPRINTZ_E_HIP, "Huff0 stage: hipMallocAsync of the descriptor workspace failed");
    const int rc = sprintz_mi355x_huf0_decompress_batch_ws(d_blocks, d_block_offsets, nchunks, d_out, d_out_offsets, d_rets, tmp, hip_stream);
    (void)hipFreeAsync(tmp, (hipStream_t)hip_stream);
    return rc;
}

}  // extern "C"

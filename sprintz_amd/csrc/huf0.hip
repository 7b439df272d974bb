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
constexpr int kRStride = 336 + 4;                // per chunk: header copy 144 | norm 32 | next 32 | fse 128; later the sorted symbols (256)
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

__device__ __forceinline__ int quad_bcast0(int v) { return __builtin_amdgcn_mov_dpp(v, 0x00, 0xf, 0xf, true); }
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ void __launch_bounds__(64) huf0_decode_kernel(const uint8_t* __restrict__ blocks, const uint64_t* __restrict__ boffs,
                                                         uint64_t nchunks, uint8_t* __restrict__ out,
                                                         const uint64_t* __restrict__ ooffs, int64_t* __restrict__ rets)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_w[16 * kWStride];
    __shared__ __attribute__((aligned(16))) uint8_t s_r[16 * kRStride];
    __shared__ uint32_t s_tab[16][17];                            // [w]: start[w] | symoff[w] << 16; [16]: running offsets are in s_run
    __shared__ uint16_t s_run[16][16];
    const int t = threadIdx.x, q = t >> 2, j = t & 3;
    const uint64_t chunk = (uint64_t)blockIdx.x * 16 + (uint64_t)q;
    const bool exists = chunk < nchunks;
    const uint64_t b0 = exists ? boffs[chunk] : 0, b1 = exists ? boffs[chunk + 1] : 0;
    const uint64_t o0 = exists ? ooffs[chunk] : 0, o1 = exists ? ooffs[chunk + 1] : 0;
    const uint8_t* const src = blocks + b0;
    uint8_t* const dst = out + o0;
    const uint64_t csize = b1 - b0, dsize = o1 - o0;
    uint8_t* const wts = s_w + q * kWStride;
    uint8_t* const scratch = s_r + q * kRStride;
    uint8_t* const sorted = scratch;                              // phase B on: the symbols by (weight, symbol)

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

    // ---- phase A: tree description -> weights (first lane of the quad)
    const uint32_t hcopy = mode == 3 ? (uint32_t)(csize < 129 ? csize : 129) : 0u;
    for (uint32_t k = 4u * (uint32_t)j; k < 144u; k += 16) {     // a dword per lane per trip (byte loads cost the address path as much)
        uint32_t v = 0;
        if (k + 4 <= hcopy) v = *(const u32_a1*)(src + k);
        else for (uint32_t b = 0; b < 4 && k + b < hcopy; b++) v |= (uint32_t)src[k + b] << (8 * b);
        *(uint32_t*)(scratch + k) = v;
    }
    wave_sync();
    uint32_t hl = 0, nsym = 0, tl = 0;
    if (mode == 3 && j == 0) {
        hl = read_stats(scratch, hcopy, wts, scratch + 144, nsym, tl);
        if (hl == 0 || hl >= csize) { hl = 0; ret = kCorrupt; }
        if (hl) {
            // ---- phase B: start[w] (first table index of weight w), symoff[w], and the symbols sorted by (weight, symbol)
            uint32_t cnt[13];
#pragma unroll
            for (int w = 0; w < 13; w++) cnt[w] = 0;
            for (uint32_t sy = 0; sy < nsym; sy++) {
                const uint32_t w = wts[sy];
#pragma unroll
                for (int ww = 1; ww < 13; ww++) cnt[ww] += (uint32_t)(w == (uint32_t)ww);
            }
            uint32_t at = 0, so = 0;
#pragma unroll
            for (int w = 1; w < 13; w++) {
                s_tab[q][w] = at | (so << 16);
                s_run[q][w] = (uint16_t)so;
                at += cnt[w] << (w - 1);
                so += cnt[w];
            }
            for (uint32_t sy = 0; sy < nsym; sy++) {
                const uint32_t w = wts[sy];
                if (w) { const uint32_t pos = s_run[q][w]; s_run[q][w] = (uint16_t)(pos + 1); sorted[pos] = (uint8_t)sy; }
            }
        }
    }
    hl = (uint32_t)quad_bcast0((int)hl);
    tl = (uint32_t)quad_bcast0((int)tl);
    wave_sync();
    const bool coded = mode == 3 && hl != 0;

    // ---- phase C: lane j decodes stream j (HUF_decompress4X1_usingDTable_internal).  Set-up per lane, then ONE
    // wave-uniform loop: the quad exchanges of the output path need every lane, streamless ones included.
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    typedef v4u __attribute__((aligned(1), may_alias)) v4u_a1;
    typedef v4u __attribute__((aligned(16), may_alias)) v4u_a16;
    bool bad = false;
    uint32_t T[13];                                               // T[k] = start[k]: thresholds of the weight search (k > tableLog: never reached)
#pragma unroll
    for (int k = 2; k < 13; k++) T[k] = (coded && (uint32_t)k <= tl) ? (s_tab[q][k] & 0xffffu) : 0xffffu;
    const uint8_t* sp = blocks;                                   // this lane's stream
    int64_t P = 0;
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
                P = 8 * (int64_t)(slen - 1) + highbit(sp[slen - 1]);
                op = dst + w0;
                left = w1 - w0;
            }
        }
    }
    const bool streaming = left > 0 || (coded && !bad);           // has a stream whose end must be checked
    const uint32_t look_shift = 32u - (tl ? tl : 1u);
    // The stream reaches the lane as 16-byte ALIGNED pieces, three of them in registers: the one that holds the
    // cursor's byte (pc0), the one below (pc1) and the one below that, in flight (pn).  A step takes at most 6
    // bytes, so the cursor leaves a piece every ~3 steps: only then is a new piece requested -- a third of the
    // requests of "16 unaligned bytes every step", which kept the address path 66 % busy -- and a piece has two
    // crossings (>= 5 steps) to arrive.  Piece k covers bytes [16 k - s_al, 16 k - s_al + 16) of the stream.
    const uint32_t s_al = (uint32_t)((uintptr_t)sp & 15u);
    const uint8_t* const sp_al = sp - s_al;                       // >= blocks: the API asks for a 16-byte aligned buffer
    auto load_piece = [&](int32_t k) -> v4u {
        if (!streaming || k < 0) return v4u{0, 0, 0, 0};
        return *(const v4u_a16*)(sp_al + 16 * (int64_t)k);
    };
    int32_t cur_k = (int32_t)(((P > 0 ? (P - 1) >> 3 : 0) + (int64_t)s_al) >> 4);
    v4u pc0 = load_piece(cur_k), pc1 = load_piece(cur_k - 1), pn = load_piece(cur_k - 2);
    // one step = 4 symbols = one dword of output
    auto step = [&](uint32_t m) -> uint32_t {
        uint64_t win = 0;
        const int64_t Pc = P;
        if (Pc > 0) {
            const uint32_t r = (uint32_t)((Pc - 1) >> 3) + s_al;  // the cursor's byte, counted from piece 0
            if ((int32_t)(r >> 4) < cur_k) {                      // left pc0: shift the pieces up, request the next one down
                pc0 = pc1;
                pc1 = pn;
                cur_k--;
                pn = load_piece(cur_k - 2);
            }
            // the 8 bytes ending at byte r, out of the 32 of (pc1 | pc0): first byte at o = 9 .. 24
            const uint32_t o = r - 7u - 16u * (uint32_t)(cur_k - 1);
            const uint32_t t3 = (o >> 2) - 2u;                    // 0 .. 4: which dword the window starts in, minus 2
            const bool b0 = (t3 & 1u) != 0, b1 = (t3 & 2u) != 0, b2 = (t3 & 4u) != 0;
            const uint32_t D2 = pc1.z, D3 = pc1.w, D4 = pc0.x, D5 = pc0.y, D6 = pc0.z, D7 = pc0.w;
            const uint32_t wa = b2 ? D6 : (b1 ? (b0 ? D5 : D4) : (b0 ? D3 : D2));
            const uint32_t wb = b2 ? D7 : (b1 ? (b0 ? D6 : D5) : (b0 ? D4 : D3));
            const uint32_t wc = b2 ? 0u : (b1 ? (b0 ? D7 : D6) : (b0 ? D5 : D4));
            const uint32_t lo = __builtin_amdgcn_alignbyte(wb, wa, o & 3u), hi = __builtin_amdgcn_alignbyte(wc, wb, o & 3u);
            win = (((uint64_t)hi << 32) | lo) << (7 - (int)((Pc - 1) & 7));
            if (Pc < 64) win &= ~0ull << (64 - (int)Pc);         // nothing before the stream's first bit
        }
        uint32_t word = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if ((uint32_t)k < m) {
                const uint32_t idx = (uint32_t)(win >> 32) >> look_shift;
                uint32_t w = 1;
#pragma unroll
                for (int kk = 2; kk < 13; kk++) w += (uint32_t)(idx >= T[kk]);
                const uint32_t e = s_tab[q][w];
                const uint32_t sym = sorted[(e >> 16) + ((idx - (e & 0xffffu)) >> (w - 1u))];
                const uint32_t nb = tl + 1u - w;
                word |= sym << (8 * k);
                win <<= nb;
                P -= (int64_t)nb;
            }
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
            wb[sN] = m ? step(m) : 0u;
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
                *(v4u_a1*)(uintptr_t)(da + 16u * part) = piece;
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

int sprintz_mi355x_huf0_decompress_batch(const void* d_blocks, const uint64_t* d_block_offsets, uint64_t nchunks, void* d_out,
                                         const uint64_t* d_out_offsets, int64_t* d_rets, void* hip_stream)
{
    if (!d_blocks || !d_block_offsets || !d_out || !d_out_offsets || ((uintptr_t)d_blocks & 15)) return sprintz::set_error(SPRINTZ_E_INVALID, "Huff0 stage: invalid argument (null pointer, alignment or size)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return sprintz::set_error(SPRINTZ_E_NO_DEVICE, "Huff0 stage: no usable HIP device (there is no CPU fallback)");
    if (nchunks == 0) return 0;
    const uint64_t grid = (nchunks + 15) / 16;
    if (grid > 0x7fffffffull) return sprintz::set_error(SPRINTZ_E_INVALID, "Huff0 stage: invalid argument (null pointer, alignment or size)");
    hipLaunchKernelGGL(huf0_decode_kernel, dim3((unsigned)grid), dim3(64), 0, (hipStream_t)hip_stream, (const uint8_t*)d_blocks,
                       d_block_offsets, nchunks, (uint8_t*)d_out, d_out_offsets, d_rets);
    return hipGetLastError() == hipSuccess ? 0 : sprintz::set_error(SPRINTZ_E_HIP, "Huff0 stage: a HIP call or kernel launch failed");
}

}  // extern "C"

// huf0.hip -- batched decoder for genuine Huff0 blocks (Yann Collet's Huff0 / FSE as shipped in
// zstd 1.4.x: HUF_compress's output = tree description + jump table + 4 bit streams), one block
// per chunk -- the entropy stage the paper applies after bit-packing (communicate/ubicomp/
// method.tex:293-297) in ITS wire format, where huf.hip is this repository's own GPU-shaped
// container.  The format is restated, with the library citations, in oracle/huf0_oracle.c;
// parity is pinned against the system libzstd's HUF_compress / HUF_decompress by the tests.
//
// A Huff0 block is 4-way parallel by construction and its code table is private to the block:
// a wave takes 16 chunks, lane = (chunk, stream).  Per chunk in LDS: the 2^tableLog-entry
// decoding table (tableLog <= 11, HUF_compress's default cap) whose 4 KB first serve as scratch
// for the FSE-compressed weights.  Phase A (first lane of each quad, serial): tree description
// -> weights.  Phase B (the quad): weights -> table, entries interleaved over its 4 lanes.
// Phase C (every lane): its stream, read from the last byte down through a 64-bit window,
// 4 symbols per refill, the next refill's 16 bytes requested a step ahead.  This is the interoperability path, not the fast one: the tables cap
// the occupancy at two waves per CU.  Measured on the headline shape (131 072 blocks of ~3.4 KB):
// 4.6 ms, of which phase A 1.9 (a quarter of the lanes, serial FSE), B 0.6, C 2.1.
#include "../../include/sprintz_mi355x.h"

#include <hip/hip_runtime.h>

#include <string>

namespace {

constexpr int kTL = 11;                          // largest table log decoded (HUF_TABLELOG_DEFAULT); the format allows 12
constexpr int kDtStride = (1 << kTL) + 8;        // u16 entries per chunk, padded off the bank stride
constexpr int kWStride = 256 + 4;
constexpr int64_t kCorrupt = SPRINTZ_E_CORRUPT;
constexpr int64_t kUnsupported = SPRINTZ_E_UNSUPPORTED;

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
// Scratch s: int16 norm[256] | u16 next[256] | u32 fse[64].  Returns header bytes, 0 if damaged.
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
        uint16_t* const next = (uint16_t*)(s + 512);
        uint32_t* const fse = (uint32_t*)(s + 1024);              // symbol | nbits << 8 | new_state << 16
        for (int k = 0; k < 256; k++) norm[k] = 0;
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
            norm[charnum++] = (int16_t)count;
            previous0 = count == 0;
            while (remaining < threshold) { nb--; threshold >>= 1; }
        }
        if (remaining != 1 || bp > bit_end || charnum == 0) return 0;
        const uint32_t max_sv = charnum - 1, hl = (bp + 7) >> 3;
        if (hl >= isize) return 0;
        // FSE_buildDTable
        const uint32_t size = 1u << tl;
        uint32_t high = size - 1;
        for (uint32_t sy = 0; sy <= max_sv; sy++) {
            if (norm[sy] == -1) { fse[high--] = sy; next[sy] = 1; }
            else next[sy] = (uint16_t)norm[sy];
        }
        {
            const uint32_t mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
            uint32_t pos = 0, placed = 0;
            for (uint32_t sy = 0; sy <= max_sv; sy++)
                for (int i = 0; i < norm[sy]; i++) {
                    if (++placed > size) return 0;
                    fse[pos] = sy;
                    pos = (pos + step) & mask;
                    while (pos > high) pos = (pos + step) & mask;
                }
            if (pos != 0) return 0;
        }
        for (uint32_t u = 0; u < size; u++) {
            const uint32_t sy = fse[u] & 0xffu, ns = next[sy]++;
            if (ns == 0 || ns >= 2 * size) return 0;
            const uint32_t nbits = tl - (uint32_t)highbit(ns);
            fse[u] = sy | (nbits << 8) | (((ns << nbits) - size) << 16);
        }
        // FSE_decompress_usingDTable: two interleaved states; the stream ends by running dry
        const uint8_t* const b = f + hl;
        const uint32_t bn = isize - hl;
        if (b[bn - 1] == 0) return 0;
        int P = 8 * (int)(bn - 1) + highbit(b[bn - 1]);
        uint32_t s1 = back_look(b, P, (int)tl); P -= (int)tl;
        uint32_t s2 = back_look(b, P, (int)tl); P -= (int)tl;
        osize = 0;
        for (;;) {
            if (osize + 2 > 255u) return 0;
            uint32_t e = fse[s1];
            weights[osize++] = (uint8_t)e;
            int nbt = (int)((e >> 8) & 0xffu);
            s1 = (e >> 16) + back_look(b, P, nbt); P -= nbt;
            if (P < 0) { weights[osize++] = (uint8_t)fse[s2]; break; }
            if (osize + 2 > 255u) return 0;
            e = fse[s2];
            weights[osize++] = (uint8_t)e;
            nbt = (int)((e >> 8) & 0xffu);
            s2 = (e >> 16) + back_look(b, P, nbt); P -= nbt;
            if (P < 0) { weights[osize++] = (uint8_t)fse[s1]; break; }
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

__global__ void __launch_bounds__(64) huf0_decode_kernel(const uint8_t* __restrict__ blocks, const uint64_t* __restrict__ boffs,
                                                         uint64_t nchunks, uint8_t* __restrict__ out,
                                                         const uint64_t* __restrict__ ooffs, int64_t* __restrict__ rets)
{
    __shared__ __attribute__((aligned(16))) uint16_t s_dt[16 * kDtStride];
    __shared__ __attribute__((aligned(16))) uint8_t s_w[16 * kWStride];
    __shared__ uint16_t s_start[16][16];
    const int t = threadIdx.x, q = t >> 2, j = t & 3;
    const uint64_t chunk = (uint64_t)blockIdx.x * 16 + (uint64_t)q;
    const bool exists = chunk < nchunks;
    const uint64_t b0 = exists ? boffs[chunk] : 0, b1 = exists ? boffs[chunk + 1] : 0;
    const uint64_t o0 = exists ? ooffs[chunk] : 0, o1 = exists ? ooffs[chunk + 1] : 0;
    const uint8_t* const src = blocks + b0;
    uint8_t* const dst = out + o0;
    const uint64_t csize = b1 - b0, dsize = o1 - o0;
    uint16_t* const dt = s_dt + q * kDtStride;
    uint8_t* const wts = s_w + q * kWStride;

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

    // ---- phase A: tree description -> weights (first lane of the quad); the table's bytes are the scratch
    uint8_t* const scratch = (uint8_t*)dt;
    const uint32_t hcopy = mode == 3 ? (uint32_t)(csize < 129 ? csize : 129) : 0u;
    for (uint32_t k = (uint32_t)j; k < 144u; k += 4) scratch[k] = k < hcopy ? src[k] : (uint8_t)0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint32_t hl = 0, nsym = 0, tl = 0;
    if (mode == 3 && j == 0) {
        hl = read_stats(scratch, hcopy, wts, scratch + 160, nsym, tl);
        if (hl == 0 || hl >= csize) { hl = 0; ret = kCorrupt; }
        else if (tl > (uint32_t)kTL) { hl = 0; ret = kUnsupported; }
        if (hl) {                                                 // HUF_readDTableX1: per weight ascending symbols, weight 1 lowest
            uint32_t cnt[13];
#pragma unroll
            for (int w = 0; w < 13; w++) cnt[w] = 0;
            for (uint32_t s = 0; s < nsym; s++) {
                const uint32_t w = wts[s];
#pragma unroll
                for (int ww = 1; ww < 13; ww++) cnt[ww] += (uint32_t)(w == (uint32_t)ww);
            }
            uint32_t at = 0;
#pragma unroll
            for (int w = 1; w < 13; w++) { s_start[q][w] = (uint16_t)at; at += cnt[w] << (w - 1); }
        }
    }
    hl = (uint32_t)quad_bcast0((int)hl);
    nsym = (uint32_t)quad_bcast0((int)nsym);
    tl = (uint32_t)quad_bcast0((int)tl);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const bool coded = mode == 3 && hl != 0;

    // ---- phase B: the decoding table, entries interleaved over the quad
    if (coded) {
        for (uint32_t s = 0; s < nsym; s++) {
            const uint32_t w = wts[s];
            if (w == 0) continue;                                 // quad-uniform
            const uint32_t len = (1u << w) >> 1, st = s_start[q][w];
            const uint16_t e = (uint16_t)(s | ((tl + 1u - w) << 8));
            for (uint32_t u = st + (uint32_t)j; u < st + len; u += 4) dt[u] = e;
            if (j == 0) s_start[q][w] = (uint16_t)(st + len);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- phase C: lane j decodes stream j (HUF_decompress4X1_usingDTable_internal)
    bool bad = false;
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
            const uint8_t* const sp = ip + so;
            uint64_t w0 = seg * (uint64_t)j;
            w0 = w0 < dsize ? w0 : dsize;
            const uint64_t w1 = j == 3 ? dsize : (w0 + seg < dsize ? w0 + seg : dsize);
            if (slen < 1 || sp[slen - 1] == 0) bad = true;
            if (!bad) {
                int64_t P = 8 * (int64_t)(slen - 1) + highbit(sp[slen - 1]);
                const uint32_t look_shift = 64u - tl;
                uint8_t* op = dst + w0;
                uint64_t left = w1 - w0;
                // 16 bytes ending at the byte that holds bit P-1 are requested one step AHEAD: four symbols
                // take at most 44 bits, so the next step's 8-byte window lies inside them (clamped to the
                // block's first byte -- at least 7 bytes precede every stream, not always 15)
                typedef uint32_t v4u __attribute__((ext_vector_type(4)));
                typedef v4u __attribute__((aligned(1), may_alias)) v4u_a1;
                const int64_t floor_b = -(int64_t)(hl + so);                 // the block's first byte, relative to sp
                auto fetch = [&](int64_t Pn, int64_t& base) -> v4u {
                    const int64_t tb = Pn > 0 ? (Pn - 1) >> 3 : 0;
                    base = tb - 15 > floor_b ? tb - 15 : floor_b;
                    return *(const v4u_a1*)(sp + base);
                };
                int64_t base = 0;
                v4u buf = fetch(P, base);
                while (left > 0) {
                    uint64_t win = 0;
                    const v4u cur = buf;
                    const int64_t cur_base = base;
                    const int64_t Pc = P;
                    if (Pc > 0) {
                        const int64_t tb = (Pc - 1) >> 3;
                        const uint32_t o = (uint32_t)(tb - 7 - cur_base);    // 0 .. 8: the window's first byte inside cur
                        const uint32_t d0 = o < 4 ? cur.x : o < 8 ? cur.y : cur.z;
                        const uint32_t d1 = o < 4 ? cur.y : o < 8 ? cur.z : cur.w;
                        const uint32_t d2 = o < 4 ? cur.z : o < 8 ? cur.w : 0u;
                        const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, o & 3u), hi = __builtin_amdgcn_alignbyte(d2, d1, o & 3u);
                        win = (((uint64_t)hi << 32) | lo) << (7 - (int)((Pc - 1) & 7));
                        if (Pc < 64) win &= ~0ull << (64 - (int)Pc);         // nothing before the stream's first bit
                    }
                    buf = fetch(Pc, base);                                   // for the NEXT step; in flight during this one's lookups
                    uint32_t word = 0;
                    const uint32_t m = left < 4 ? (uint32_t)left : 4u;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        if ((uint32_t)k < m) {
                            const uint32_t e = dt[win >> look_shift];
                            const uint32_t nb = e >> 8;
                            word |= (e & 0xffu) << (8 * k);
                            win <<= nb;
                            P -= (int64_t)nb;
                        }
                    }
                    if (m == 4) *(u32_a1*)op = word;
                    else for (uint32_t k = 0; k < m; k++) op[k] = (uint8_t)(word >> (8 * k));
                    op += m;
                    left -= m;
                    if (P < -64) break;                                      // damaged: ran far past the start
                }
                if (P != 0) bad = true;                                      // every stream ends exactly (BIT_endOfDStream)
            }
        }
    }
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
    if (!d_blocks || !d_block_offsets || !d_out || !d_out_offsets) return SPRINTZ_E_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return SPRINTZ_E_NO_DEVICE;
    if (nchunks == 0) return 0;
    const uint64_t grid = (nchunks + 15) / 16;
    if (grid > 0x7fffffffull) return SPRINTZ_E_INVALID;
    hipLaunchKernelGGL(huf0_decode_kernel, dim3((unsigned)grid), dim3(64), 0, (hipStream_t)hip_stream, (const uint8_t*)d_blocks,
                       d_block_offsets, nchunks, (uint8_t*)d_out, d_out_offsets, d_rets);
    return hipGetLastError() == hipSuccess ? 0 : SPRINTZ_E_HIP;
}

}  // extern "C"

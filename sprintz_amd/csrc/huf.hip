// huf.hip -- optional Huffman stage over the bit-packed Sprintz streams (bytes as
// symbols, applied after bit-packing, as the paper does with Huff0:
// communicate/ubicomp/method.tex:293-297).
//
// Parity status: UNPINNED against the reference -- dblalock/sprintz ships no Huffman
// coder.  The container format is this repository's own and is
// specified in oracle/huf_oracle.c (segments of 64 chunks share a code table, 4
// byte-aligned sub-streams per chunk, 11-bit length limit); the kernels here are
// bit-exact with that CPU oracle.  Everything upstream of this stage (the Sprintz
// streams) stays bit-exact with the reference.
#include "../../include/sprintz_mi355x.h"

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <string>

#include "launch.h"

using namespace sprintz;

namespace {

constexpr int LMAX = 11;
constexpr int SEG = 64;                 // chunks per segment = 256 threads / 4 sub-streams

typedef uint32_t __attribute__((aligned(1), may_alias)) u32_any;

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((aligned(4), may_alias)) u32x4_a4;
typedef u32x4 __attribute__((aligned(1), may_alias)) u32x4_a1;

// wave-wide maximum on the DPP network (quad swaps, the row mirrors, row_bcast15 / 31, lane 63 read back): ~10 dependent vector
// instructions instead of six ds_bpermute round trips.  (The length-limit repair of huf_build_kernel calls the 64-bit form once per
// round, up to a few hundred rounds a segment: with __shfl_xor on 64 bits -- twelve LDS-crossbar trips a round -- it was most of the kernel.)
__device__ __forceinline__ uint32_t wave_max_u32_dpp(uint32_t v)
{
    auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xf, 0xf, false));   // row_half_mirror
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xf, 0xf, false));   // row_mirror: every lane of a row holds the row's
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xa, 0xf, false));   // row_bcast15 into rows 1 and 3
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xc, 0xf, false));   // row_bcast31 into rows 2 and 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v)
{
    const uint32_t hi = wave_max_u32_dpp((uint32_t)(v >> 32));
    const uint32_t lo = wave_max_u32_dpp((uint32_t)(v >> 32) == hi ? (uint32_t)v : 0u);
    return ((uint64_t)hi << 32) | lo;
}

#ifdef HUF_BUILD_TIMING                         // experiment builds: where a segment's workgroup spends its time (block 0, thread 0; 10 ns ticks)
__device__ uint64_t g_build_ts[16];
#define BUILD_TS(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_build_ts[k] = wall_clock64(); } while (0)
#else
#define BUILD_TS(k) do {} while (0)
#endif

// ---------------------------------------------------------------- K1: histogram + code lengths + codes
// One workgroup per segment.  The histogram and the sort run on all 256 threads, the
// two-queue merge (inherently serial, 255 steps) on thread 0, the length-limit repair --
// up to a few hundred "find the best symbol, move it one level" rounds -- as a wave-wide
// arg-max per round instead of a 256-entry scan per round.
__global__ void __launch_bounds__(256) huf_build_kernel(const uint8_t* dense, const uint64_t* offsets, const uint32_t* sizes,
                                                        uint64_t nchunks, uint32_t* enc_tables, uint8_t* tables)
{
    constexpr int kHistCopies = 16;                    // (round 4: one per wave -- 64 lanes on <= 256 counters, the hot byte values collide)
    __shared__ uint32_t hist4[kHistCopies][256];
    __shared__ uint32_t hist[256];
    __shared__ uint32_t lcnt[256];      // leaves sorted by (count, symbol): count
    __shared__ uint16_t lsym[256];      //                                   symbol
    __shared__ uint32_t w[511];
    __shared__ uint16_t parent[511];
    __shared__ uint8_t lens[256];
    __shared__ uint32_t first[LMAX + 2];
    __shared__ uint32_t s_kraft;
    __shared__ int s_z;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const uint64_t seg = blockIdx.x;
    const uint64_t c0 = seg * SEG, c1 = (c0 + SEG < nchunks) ? c0 + SEG : nchunks;

    BUILD_TS(0);
    for (int k = 0; k < kHistCopies; k++) hist4[k][t] = 0;
    lens[t] = 0;
    if (t == 0) s_kraft = 0;
    __syncthreads();
    uint32_t* const hw = hist4[t & (kHistCopies - 1)];
    // chunk geometry once, coalesced; then two chunks' first pieces are in flight together (a
    // chunk of the headline shape is < 256 pieces: one trip per thread, and the loop was two
    // dependent memory latencies per chunk)
    __shared__ uint64_t s_off[SEG];
    __shared__ uint32_t s_sz[SEG];
    if (t < SEG) {
        const bool in = c0 + t < c1;
        s_off[t] = in ? offsets[c0 + t] : 0;
        s_sz[t] = in ? sizes[c0 + t] : 0;
    }
    __syncthreads();
    auto tally = [&](const u32x4& x) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t d = k == 0 ? x.x : k == 1 ? x.y : k == 2 ? x.z : x.w;
            atomicAdd(&hw[d & 255], 1u);
            atomicAdd(&hw[(d >> 8) & 255], 1u);
            atomicAdd(&hw[(d >> 16) & 255], 1u);
            atomicAdd(&hw[d >> 24], 1u);
        }
    };
    const int nc = (int)(c1 - c0);
    // eight chunks' first pieces in flight together (round 4: two -- the loop was 32 dependent memory latencies a segment, ~80 of the
    // kernel's 110 us; a chunk of the headline shape is < 256 pieces: one trip per thread)
    constexpr int kU = 8;
    for (int cc = 0; cc < nc; cc += kU) {
        u32x4 x[kU];
        bool h[kU];
#pragma unroll
        for (int q = 0; q < kU; q++) {
            const bool in = cc + q < nc;
            const uint32_t np = in ? s_sz[cc + q] >> 4 : 0u;
            h[q] = (uint32_t)t < np;
            x[q] = u32x4{0, 0, 0, 0};
            if (h[q]) x[q] = *(const u32x4_a1*)(dense + s_off[cc + q] + (size_t)t * 16);
        }
#pragma unroll
        for (int q = 0; q < kU; q++) if (h[q]) tally(x[q]);
#pragma unroll 1
        for (int q = 0; q < kU; q++) {
            if (cc + q >= nc) break;
            const uint8_t* const p0 = dense + s_off[cc + q];
            const uint32_t n0 = s_sz[cc + q], np0 = n0 >> 4;
            for (uint32_t i = t + 256; i < np0; i += 256) { const u32x4 xx = *(const u32x4_a1*)(p0 + (size_t)i * 16); tally(xx); }
            for (uint32_t i = (np0 << 4) + t; i < n0; i += 256) atomicAdd(&hw[p0[i]], 1u);
        }
    }
    __syncthreads();
    {
        uint32_t hsum = 0;
#pragma unroll
        for (int k = 0; k < kHistCopies; k++) hsum += hist4[k][t];
        hist[t] = hsum;
    }
    __syncthreads();

    BUILD_TS(1);
    {   // rank sort by (count, symbol)
        const uint32_t cs = hist[t];
        int rank = 0;
        for (int u = 0; u < 256; u++) {
            const uint32_t ct = hist[u];
            rank += (ct < cs || (ct == cs && u < t)) ? 1 : 0;
        }
        lcnt[rank] = cs;
        lsym[rank] = (uint16_t)t;
    }
    __syncthreads();
    BUILD_TS(2);

    {   // zero-count symbols sort first: their number, and the leaves' weights into the tree array, by everybody (round 4: thread 0
        // walked both -- 20 of the merge phase's 57 us)
        const int zc = __syncthreads_count(hist[t] == 0);
        if (t == 0) s_z = zc;
        if (t < 256 - zc) w[t] = lcnt[zc + t];
        __syncthreads();
    }
    if (t == 0) {
        const int z = s_z;
        const int nz = 256 - z;
        const uint32_t* lc = lcnt + z;
        if (nz >= 2) {
            // two-queue Huffman (oracle/huf_oracle.c: huf_oracle_lengths); (branch-free picks measured slower: 75 against 57 us)
            // (round 5: the two queues' front weights ride in registers, the leaf queue's two ahead -- the walk was six dependent LDS
            //  reads a merge, 255 merges; the picks and the tree are the same)
            int ql = 0, qi = nz, next = nz;
            uint32_t lw0 = lc[0], lw1 = nz > 1 ? lc[1] : 0u;         // w[ql], w[ql + 1]
            uint32_t iw0 = 0, iw1 = 0;                                // w[qi], w[qi + 1] where those exist (qi < next, qi + 1 < next)
            for (int m = 0; m < nz - 1; m++) {
                int pick[2];
                uint32_t sum = 0;
                for (int k = 0; k < 2; k++) {
                    const bool has_l = ql < nz, has_i = qi < next;
                    if (has_l && (!has_i || lw0 <= iw0)) {
                        pick[k] = ql++;
                        sum += lw0;
                        lw0 = lw1;
                        lw1 = ql + 1 < nz ? lc[ql + 1] : 0u;
                    } else {
                        pick[k] = qi++;
                        sum += iw0;
                        iw0 = iw1;
                        iw1 = qi + 1 < next ? w[qi + 1] : 0u;
                    }
                }
                w[next] = sum;
                if (qi == next) iw0 = sum;
                else if (qi + 1 == next) iw1 = sum;
                parent[pick[0]] = (uint16_t)next;
                parent[pick[1]] = (uint16_t)next;
                next++;
            }
        }
    }
    __syncthreads();
    BUILD_TS(3);
    const int z = s_z;
    const int nz = 256 - z;
    if (nz == 1) {
        if (t == 0) lens[lsym[z]] = 1;
    } else if (nz >= 2) {
        // leaf t of the sorted order: depth = steps to the root, clamped to LMAX
        int myl = 0;
        if (t < nz) {
            const int root = 2 * nz - 2;
            int node = t, d = 0;
            while (node != root) { node = parent[node]; d++; }
            myl = d > LMAX ? LMAX : d;
            atomicAdd(&s_kraft, 1u << (LMAX - myl));
            lens[lsym[z + t]] = (uint8_t)myl;
        }
        __syncthreads();
        BUILD_TS(4);
        if (t < 64 && s_kraft != (1u << LMAX)) {             // wave 0 repairs; 4 leaves per lane
            uint32_t kraft = s_kraft;
            uint32_t cn[4], sy[4], ln[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int i = lane + 64 * e;
                const bool ok = i < nz;
                cn[e] = ok ? lcnt[z + i] : 0u;
                sy[e] = ok ? lsym[z + i] : 0u;
                ln[e] = ok ? lens[sy[e]] : 0u;
            }
            // too many codes for LMAX bits: lengthen the deepest symbol below LMAX (ties:
            // smallest count, then largest symbol) until the code fits
            while (kraft > (1u << LMAX)) {
                uint64_t key[4], best = 0;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    key[e] = (ln[e] != 0 && ln[e] < (uint32_t)LMAX)
                                 ? ((uint64_t)ln[e] << 48) | ((uint64_t)(0xffffffffu - cn[e]) << 16) | sy[e] : 0;
                    best = key[e] > best ? key[e] : best;
                }
                best = wave_max_u64(best);
#pragma unroll
                for (int e = 0; e < 4; e++) ln[e] += (key[e] == best) ? 1u : 0u;
                kraft -= 1u << (LMAX - (uint32_t)(best >> 48) - 1);
            }
            // room left: shorten the most frequent symbol that still fits (ties: smallest symbol)
            for (;;) {
                uint64_t key[4], best = 0;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    key[e] = (ln[e] > 1 && kraft + (1u << (LMAX - ln[e])) <= (1u << LMAX))
                                 ? ((uint64_t)cn[e] << 16) | ((uint64_t)(255u - sy[e]) << 4) | ln[e] : 0;
                    best = key[e] > best ? key[e] : best;
                }
                best = wave_max_u64(best);
                if (best == 0) break;
#pragma unroll
                for (int e = 0; e < 4; e++) ln[e] -= (key[e] == best) ? 1u : 0u;
                kraft += 1u << (LMAX - (uint32_t)(best & 15));
            }
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (lane + 64 * e < nz) lens[sy[e]] = (uint8_t)ln[e];
        }
    }
    __syncthreads();
    BUILD_TS(5);
    // first canonical code of every length: the symbols per length counted by all threads (round 4: thread 0 walked the 256 lengths
    // into a dynamically indexed local array)
    __shared__ uint32_t s_count[LMAX + 2];
    if (t < LMAX + 2) s_count[t] = 0;
    __syncthreads();
    atomicAdd(&s_count[lens[t] <= LMAX + 1 ? lens[t] : LMAX + 1], 1u);
    __syncthreads();
    if (t == 0) {
        uint32_t code = 0, prev = 0;                         // (count[0] does not count)
        first[0] = 0;
        for (int l = 1; l <= LMAX; l++) { code = (code + prev) << 1; first[l] = code; prev = s_count[l]; }
    }
    __syncthreads();

    {
        const int l = lens[t];
        uint32_t e = 0;
        if (l) {
            int before = 0;
            for (int u = 0; u < t; u++) before += lens[u] == l;
            const uint32_t c = first[l] + (uint32_t)before;
            e = (__brev(c) >> (32 - l)) | ((uint32_t)l << 16);
        }
        enc_tables[seg * 256 + t] = e;
    }
    if (t < 128) tables[seg * 128 + t] = (uint8_t)(lens[2 * t] | (lens[2 * t + 1] << 4));
    BUILD_TS(6);
}

// sub-stream j of an n-symbol chunk: [a, b)
__device__ __forceinline__ void sub_range(uint32_t n, int j, uint32_t& a, uint32_t& b)
{
    const uint32_t q = (n + 3u) >> 2;
    a = (uint32_t)j * q < n ? (uint32_t)j * q : n;
    b = (a + q < n) ? a + q : n;
}

// ---------------------------------------------------------------- K2: encoded sizes
// one workgroup per segment.  The chunks are read cooperatively (a 16-byte piece per
// thread, consecutive threads on consecutive pieces) and the code lengths summed into one
// LDS counter per (chunk, sub-stream).
__global__ void __launch_bounds__(256) huf_size_kernel(const uint8_t* dense, const uint64_t* offsets, const uint32_t* sizes,
                                                       uint64_t nchunks, const uint32_t* enc_tables, uint32_t* rec_sizes, uint64_t* meta)
{
    __shared__ uint32_t enc[256];
    __shared__ uint32_t cnt[SEG * 4];
    const uint64_t seg = blockIdx.x;
    const uint32_t t = threadIdx.x;
    enc[t] = enc_tables[seg * 256 + t] >> 16;                               // code lengths only
    cnt[t] = 0;
    __syncthreads();
    const uint64_t c0 = seg * SEG, c1 = (c0 + SEG < nchunks) ? c0 + SEG : nchunks;
    __shared__ uint64_t s_off[SEG];
    __shared__ uint32_t s_sz[SEG];
    if (t < SEG) {                                                          // chunk geometry once, coalesced
        const bool in = c0 + t < c1;
        s_off[t] = in ? offsets[c0 + t] : 0;
        s_sz[t] = in ? sizes[c0 + t] : 0;
    }
    __syncthreads();
    // piece i of local chunk cc: x already loaded when the piece lies inside one sub-stream
    auto account = [&](int cc, uint32_t i, bool whole, const u32x4& x) {
        const uint32_t n = s_sz[cc];
        const uint8_t* s = dense + s_off[cc];
        const uint32_t q = (n + 3u) >> 2;
        uint32_t* const my = cnt + cc * 4;
        auto stream_of = [&](uint32_t pos) { return (uint32_t)(pos >= q) + (uint32_t)(pos >= 2 * q) + (uint32_t)(pos >= 3 * q); };
        const uint32_t pos0 = i * 16;
        if (whole) {
            uint32_t bits = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t d = k == 0 ? x.x : k == 1 ? x.y : k == 2 ? x.z : x.w;
                bits += enc[d & 255] + enc[(d >> 8) & 255] + enc[(d >> 16) & 255] + enc[d >> 24];
            }
            atomicAdd(&my[stream_of(pos0)], bits);
        } else {
            const uint32_t end = pos0 + 16 < n ? pos0 + 16 : n;
            for (uint32_t pos = pos0; pos < end; pos++) atomicAdd(&my[stream_of(pos)], enc[s[pos]]);
        }
    };
    auto is_whole = [&](int cc, uint32_t i) {
        const uint32_t n = s_sz[cc], q = (n + 3u) >> 2, pos0 = i * 16;
        auto stream_of = [&](uint32_t pos) { return (uint32_t)(pos >= q) + (uint32_t)(pos >= 2 * q) + (uint32_t)(pos >= 3 * q); };
        return pos0 + 16 <= n && stream_of(pos0) == stream_of(pos0 + 15);
    };
    const int nc = (int)(c1 - c0);
    for (int cc = 0; cc < nc; cc += 2) {                                    // two chunks' first pieces in flight together
        const int cd = cc + 1 < nc ? cc + 1 : cc;
        const uint32_t np0 = (s_sz[cc] + 15u) >> 4, np1 = cd != cc ? (s_sz[cd] + 15u) >> 4 : 0u;
        const bool h0 = t < np0, h1 = t < np1;
        const bool w0 = h0 && is_whole(cc, t), w1 = h1 && is_whole(cd, t);
        u32x4 x0 = {0, 0, 0, 0}, x1 = {0, 0, 0, 0};
        if (w0) x0 = *(const u32x4_a1*)(dense + s_off[cc] + (size_t)t * 16);
        if (w1) x1 = *(const u32x4_a1*)(dense + s_off[cd] + (size_t)t * 16);
        if (h0) account(cc, t, w0, x0);
        if (h1) account(cd, t, w1, x1);
        for (uint32_t i = t + 256; i < np0; i += 256) {
            const bool w = is_whole(cc, i);
            u32x4 x = {0, 0, 0, 0};
            if (w) x = *(const u32x4_a1*)(dense + s_off[cc] + (size_t)i * 16);
            account(cc, i, w, x);
        }
        for (uint32_t i = t + 256; i < np1; i += 256) {
            const bool w = is_whole(cd, i);
            u32x4 x = {0, 0, 0, 0};
            if (w) x = *(const u32x4_a1*)(dense + s_off[cd] + (size_t)i * 16);
            account(cd, i, w, x);
        }
    }
    __syncthreads();
    const uint64_t c = seg * SEG + (t >> 2);
    const int j = t & 3;
    uint32_t n = 0, sz = 0;
    if (c < nchunks) {
        n = s_sz[t >> 2];
        sz = (cnt[t] + 7u) >> 3;
    }
    // the four sizes of a chunk sit in one quad
    const int q0 = (int)(threadIdx.x & 63u & ~3u);
    const uint32_t s0 = __shfl(sz, q0 + 0), s1 = __shfl(sz, q0 + 1), s2 = __shfl(sz, q0 + 2), s3 = __shfl(sz, q0 + 3);
    if (c < nchunks && j == 0) {
        const uint64_t encb = (uint64_t)s0 + s1 + s2 + s3;
        const bool stored = (12 + encb >= 4 + (uint64_t)n) || s0 > 0xffffu || s1 > 0xffffu || s2 > 0xffffu;
        rec_sizes[c] = stored ? 4u + n : 12u + (uint32_t)encb;
        meta[c] = (uint64_t)(s0 & 0xffffu) | ((uint64_t)(s1 & 0xffffu) << 16) | ((uint64_t)(s2 & 0xffffu) << 32) | ((uint64_t)stored << 48);
    }
}

// ---------------------------------------------------------------- K3: encode
// One lane per sub-stream.  A lane-per-byte-stream kernel touches every 128-byte line 32+
// times, far apart in time, with 64K such lines open per XCD -- so both sides move in
// 64-byte bursts: the source bytes are read four 16-byte loads at a time, one burst ahead;
// the coded dwords collect in a 64-byte window in LDS that is stored when full.
__global__ void __launch_bounds__(256) huf_encode_kernel(const uint8_t* dense, const uint64_t* offsets, const uint32_t* sizes,
                                                         uint64_t nchunks, const uint32_t* enc_tables, const uint64_t* meta,
                                                         uint8_t* huf, const uint64_t* huf_offsets)
{
    __shared__ uint32_t enc[256];
    __shared__ uint32_t wbuf[16 * 256];
    const uint64_t seg = blockIdx.x;
    enc[threadIdx.x] = enc_tables[seg * 256 + threadIdx.x];
    __syncthreads();
    const uint64_t c = seg * SEG + (threadIdx.x >> 2);
    const int j = threadIdx.x & 3;
    if (c >= nchunks) return;
    const uint32_t n = sizes[c];
    const uint64_t soff = offsets[c];
    const uint8_t* s = dense + soff;
    const uint64_t roff = huf_offsets[c];
    uint8_t* o = huf + roff;
    const uint64_t m = meta[c];
    const bool stored = (m >> 48) & 1;
    if (j == 0) {
        *(uint32_t*)o = n | (stored ? 0x80000000u : 0u);                    // records are 4-byte aligned
        if (!stored) { ((uint32_t*)o)[1] = (uint32_t)m; ((uint32_t*)o)[2] = (uint32_t)(m >> 32) & 0xffffu; }
    }
    if (stored) {
        const uint32_t nw = n >> 2;
        for (uint32_t i = j; i < nw; i += 4) ((uint32_t*)(o + 4))[i] = ((const u32_any*)s)[i];
        for (uint32_t i = (nw << 2) + j; i < n; i += 4) o[4 + i] = s[i];
        return;
    }
    const uint32_t sz0 = (uint32_t)m & 0xffffu, sz1 = (uint32_t)(m >> 16) & 0xffffu, sz2 = (uint32_t)(m >> 32) & 0xffffu;
    const uint64_t poff = roff + 12 + (j > 0 ? sz0 : 0u) + (j > 1 ? sz1 : 0u) + (j > 2 ? sz2 : 0u);   // container offset of the sub-stream
    uint32_t a, b;
    sub_range(n, j, a, b);

    // Output window = container bytes [wbase, wbase+64), wbase a multiple of 64, kept in LDS as
    // wbuf[dword][lane] (a lane's dwords sit in its own bank); a full window leaves as four
    // 16-byte stores back to back, only the first and last window of a sub-stream (shared with
    // its neighbours) are written piecemeal.  Same reasoning as in the decoder: touch memory in
    // 64-byte bursts, or every 16-byte access pays for a whole cache line.
    uint32_t* const wb = wbuf + threadIdx.x;
    uint64_t wbase = poff & ~(uint64_t)63;
    uint32_t wk = (uint32_t)(poff & 63) >> 2;                               // dword of the window the accumulator drains into
    uint32_t wfirst = (uint32_t)(poff & 63);                                // first window: bytes below this belong to the neighbour
    uint64_t acc = 0;
    uint32_t nbits = (uint32_t)(poff & 3) * 8;                              // the accumulator starts inside a dword
    auto store_range = [&](uint32_t lo, uint32_t hi) {                      // window bytes [lo, hi) -> container
        uint8_t* const dst = huf + wbase;
#pragma unroll
        for (uint32_t m = 0; m < 16; m++) {
            if (4 * m + 4 <= lo || 4 * m >= hi) continue;
            const uint32_t d = wb[m * 256];
            if (4 * m >= lo && 4 * m + 4 <= hi) {
                *(uint32_t*)(dst + 4 * m) = d;                              // container and window are 4-byte aligned
            } else {
                for (uint32_t k = 0; k < 4; k++)
                    if (4 * m + k >= lo && 4 * m + k < hi) dst[4 * m + k] = (uint8_t)(d >> (8 * k));
            }
        }
    };
    auto flush = [&]() {                                                    // low 32 bits of acc -> window
        wb[wk * 256] = (uint32_t)acc;
        acc >>= 32;
        nbits -= 32;
        if (++wk == 16) {
            if (wfirst) {
                store_range(wfirst, 64);
            } else {
                u32x4 p[4];
#pragma unroll
                for (int q = 0; q < 4; q++) p[q] = u32x4{wb[(4 * q) * 256], wb[(4 * q + 1) * 256], wb[(4 * q + 2) * 256], wb[(4 * q + 3) * 256]};
#pragma unroll
                for (int q = 0; q < 4; q++) *(u32x4_a4*)(huf + wbase + 16 * q) = p[q];
            }
            wfirst = 0;
            wbase += 64;
            wk = 0;
        }
    };
    auto put = [&](uint32_t sym) {
        const uint32_t e = enc[sym];
        acc |= (uint64_t)(e & 0xffffu) << nbits;
        nbits += e >> 16;
    };
    auto put16 = [&](const u32x4& v) {                                      // 16 source bytes
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t d = q == 0 ? v.x : q == 1 ? v.y : q == 2 ? v.z : v.w;
            put(d & 255u);
            put((d >> 8) & 255u);
            if (nbits >= 32) flush();
            put((d >> 16) & 255u);
            put(d >> 24);
            if (nbits >= 32) flush();
        }
    };

    // source: bytes up to a 16-byte boundary, 16-byte steps up to a 64-byte boundary, 64-byte
    // bursts (four loads back to back, one burst ahead), and the same in reverse at the end
    uint32_t i = a;
    {
        uint32_t pro = (16u - (uint32_t)((soff + a) & 15)) & 15u;
        if (pro > b - a) pro = b - a;
        for (uint32_t k = 0; k < pro; k++) {
            put(s[i++]);
            if (nbits >= 32) flush();
        }
    }
    while (((soff + i) & 63) != 0 && i + 16 <= b) {
        const u32x4 v = *(const u32x4_a1*)(s + i);
        put16(v);
        i += 16;
    }
    if (i + 64 <= b) {
        u32x4 nxt[4];
#pragma unroll
        for (int q = 0; q < 4; q++) nxt[q] = *(const u32x4_a1*)(s + i + 16 * q);
        while (i + 64 <= b) {
            u32x4 cur[4];
#pragma unroll
            for (int q = 0; q < 4; q++) cur[q] = nxt[q];
            i += 64;
            if (i + 64 <= b) {
#pragma unroll
                for (int q = 0; q < 4; q++) nxt[q] = *(const u32x4_a1*)(s + i + 16 * q);
            }
#pragma unroll
            for (int q = 0; q < 4; q++) put16(cur[q]);
        }
    }
    while (i + 16 <= b) {
        const u32x4 v = *(const u32x4_a1*)(s + i);
        put16(v);
        i += 16;
    }
    for (; i < b; i++) {
        put(s[i]);
        if (nbits >= 32) flush();
    }
    // close: the rest of the accumulator (zero padded to a byte), then the partial window
    const uint32_t endb = wk * 4 + ((nbits + 7u) >> 3);                     // bytes of the window in use (<= 64 + 3... wk < 16 here)
    if (nbits > 0) wb[wk * 256] = (uint32_t)acc;
    const uint32_t hi = endb < 64u ? endb : 64u;
    if (hi > wfirst) store_range(wfirst, hi);
}

// ---------------------------------------------------------------- K4: raw sizes from the record headers
// A record is only believed if it lies inside the container and its own header fits in it:
// nothing below reads or writes outside the buffers on damaged input.
__device__ __forceinline__ bool record_ok(const uint8_t* huf, const uint64_t* huf_offsets, uint64_t nchunks, uint64_t c, uint32_t& n)
{
    const uint64_t total = huf_offsets[nchunks];
    const uint64_t roff = huf_offsets[c], rend = huf_offsets[c + 1];
    n = 0;
    if ((roff & 3) || rend > total || roff + 4 > rend) return false;
    const uint32_t hdr = *(const uint32_t*)(huf + roff);
    const uint32_t nn = hdr & 0x7fffffffu;
    const uint64_t len = rend - roff;
    if (hdr >> 31) {
        if (4 + (uint64_t)nn > len) return false;
    } else {
        if (len < 12) return false;
        const uint32_t h0 = ((const uint32_t*)(huf + roff))[1], h1 = ((const uint32_t*)(huf + roff))[2];
        if (12 + (uint64_t)(h0 & 0xffffu) + (h0 >> 16) + (h1 & 0xffffu) > len) return false;
        if ((uint64_t)nn > 8 * len) return false;              // every symbol costs at least one bit
    }
    n = nn;
    return true;
}

__global__ void __launch_bounds__(256) huf_rawsize_kernel(const uint8_t* huf, const uint64_t* huf_offsets, uint64_t nchunks, uint32_t* sizes)
{
    const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= nchunks) return;
    uint32_t n;
    record_ok(huf, huf_offsets, nchunks, c, n);                 // a rejected record decodes to an empty stream
    sizes[c] = n;
}

// LDS hand-off between lanes of one wavefront (DS ops of a wave execute in issue order)
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------- K5: decode
__global__ void __launch_bounds__(256) huf_decode_kernel(const uint8_t* huf, const uint64_t* huf_offsets, const uint8_t* tables,
                                                         uint64_t nchunks, uint8_t* dense, const uint64_t* offsets,
                                                         uint64_t dense_capacity, int64_t* rets)
{
    // 32 KB ring + 4 KB table: four workgroups per CU.  The ring holds, per lane, the next
    // 128 coded bytes of its sub-stream as 32 dwords at ring[d][lane] -- dword d of every
    // lane sits in the lane's own bank, so per-lane cursors never conflict.  The small
    // arrays of the table build live in the (not yet used) ring.
    __shared__ uint32_t ring[32 * 256];
    __shared__ uint16_t dtab[1 << LMAX];                                    // sym | len << 8
    uint8_t* const lens = (uint8_t*)ring;
    uint32_t* const count = ring + 64;
    uint32_t* const first = ring + 64 + LMAX + 2;
    const uint64_t seg = blockIdx.x;
    const int t = threadIdx.x;
    {
        const uint8_t nib = tables[seg * 128 + (t >> 1)];
        lens[t] = (t & 1) ? (nib >> 4) : (nib & 15);
    }
    if (t < LMAX + 2) count[t] = 0;
    for (int k = t; k < (1 << LMAX); k += 256) dtab[k] = 0;
    __syncthreads();
    const int l = lens[t];
    if (l) atomicAdd(&count[l], 1u);
    __syncthreads();
    if (t == 0) {
        uint32_t code = 0;
        first[0] = 0;
        uint32_t prev = 0;
        for (int k = 1; k <= LMAX; k++) { code = (code + prev) << 1; first[k] = code; prev = count[k]; }
    }
    __syncthreads();
    if (l) {
        int before = 0;
        for (int u = 0; u < t; u++) before += lens[u] == l;
        const uint32_t cde = first[l] + (uint32_t)before;
        const uint32_t rev = __brev(cde) >> (32 - l);
        for (uint32_t k = rev; k < (1u << LMAX); k += 1u << l) dtab[k] = (uint16_t)(t | (l << 8));
    }
    __syncthreads();

    const uint64_t c = seg * SEG + (t >> 2);
    const int j = t & 3;
    if (c >= nchunks) return;
    const uint64_t roff = huf_offsets[c];
    const uint8_t* r = huf + roff;
    uint32_t n;
    const uint64_t ooff = offsets[c];
    const bool ok = record_ok(huf, huf_offsets, nchunks, c, n) && ooff + n <= dense_capacity;
    if (j == 0 && rets) rets[c] = ok ? (int64_t)n : (int64_t)SPRINTZ_E_CORRUPT;
    if (!ok) return;
    const uint32_t hdr = *(const uint32_t*)r;
    uint8_t* o = dense + ooff;
    if (hdr >> 31) {
        const uint32_t nw = n >> 2;
        for (uint32_t i = j; i < nw; i += 4) ((u32_any*)o)[i] = ((const uint32_t*)(r + 4))[i];
        for (uint32_t i = (nw << 2) + j; i < n; i += 4) o[i] = r[4 + i];
        return;
    }
    const uint32_t h0 = ((const uint32_t*)r)[1], h1 = ((const uint32_t*)r)[2];
    const uint32_t sz0 = h0 & 0xffffu, sz1 = h0 >> 16, sz2 = h1 & 0xffffu;
    const uint64_t poff = roff + 12 + (j > 0 ? sz0 : 0u) + (j > 1 ? sz1 : 0u) + (j > 2 ? sz2 : 0u);
    uint32_t a, b;
    sub_range(n, j, a, b);

    // HBM traffic.  524 288 lanes each walking their own ~0.9 KB of input and output
    // keep ~17 MB of cache lines open per XCD against a 4 MB L2: with 16-byte accesses
    // every 128-byte line was fetched 8 times and written back in 32-byte fragments
    // (measured: 4.7 GB read, 1.0 GB written for 0.44 + 0.47 GB of payload).  So a lane
    // touches memory in 64-byte bursts only: input as 64-byte pieces (four dwordx4 loads
    // back to back), output as 64 decoded symbols collected in registers and stored as
    // four dwordx4.
    //
    // Every lane decodes one symbol per step, in lockstep, so the refill is on a fixed
    // cadence: every 16 symbols (<= 176 bits consumed; a piece is 512) the piece requested
    // at the previous refill point is parked in the lane's LDS ring, and the next one is
    // requested if a slot is free.  The four loads are issued UNCONDITIONALLY (a lane with
    // no free slot reads one hot line instead) so that the compiler can count the
    // outstanding VMEM operations and wait for exactly the ones it needs.  The ring has two
    // slots: when the cursor leaves one, the other is full (>= 512 bits) and the refill
    // lands within two periods (<= 352 bits).
    // 16-byte loads past the last one that holds container bytes are clamped to it: what
    // they would deliver is never part of a valid symbol.
    const uint64_t total = huf_offsets[nchunks];
    const uint64_t last16 = (total - 1) >> 4;
    const uint64_t piece0 = poff >> 6;
    struct Piece { u32x4 v[4]; };
    const uint8_t* const idle = tables + seg * 128;                         // what a lane with no free slot reads instead (one hot line)
    auto load_piece = [&](uint32_t k, bool wanted) -> Piece {
        Piece pc;
        const uint64_t q0 = (piece0 + k) << 2;
#pragma unroll
        for (int m = 0; m < 4; m++) {
            uint64_t q = q0 + m;
            q = q < last16 ? q : last16;
            const uint8_t* src = wanted ? huf + (q << 4) : idle;
            pc.v[m] = *(const u32x4_a4*)src;
        }
        return pc;
    };
    uint32_t* const my = ring + t;
    auto park = [&](uint32_t k, const Piece& pc) {
        uint32_t* q = my + ((k & 1u) << 12);
#pragma unroll
        for (int m = 0; m < 4; m++) {
            q[(4 * m + 0) * 256] = pc.v[m].x;
            q[(4 * m + 1) * 256] = pc.v[m].y;
            q[(4 * m + 2) * 256] = pc.v[m].z;
            q[(4 * m + 3) * 256] = pc.v[m].w;
        }
    };
    park(0, load_piece(0, true));
    park(1, load_piece(1, true));
    uint32_t fpiece = 2;                                                    // pieces parked so far
    Piece pend = {};
    bool have_pend = false;                                                 // pend holds piece `fpiece`
    uint32_t bp = (uint32_t)(poff & 63) * 8;                                // bit cursor, from the start of piece0
    auto refill = [&]() {
        if (have_pend) {
            park(fpiece, pend);
            fpiece++;
        }
        have_pend = fpiece - (bp >> 9) < 2u;
        pend = load_piece(fpiece, have_pend);
    };
    auto window = [&]() -> uint32_t {                                       // the 32 bits at the cursor
        const uint32_t d0 = my[((bp >> 5) & 31u) << 8];
        const uint32_t d1 = my[(((bp >> 5) + 1u) & 31u) << 8];
        return __builtin_amdgcn_alignbit(d1, d0, bp);                       // shift = bp[4:0]
    };
    auto sym = [&](uint32_t& w) -> uint32_t {
        const uint32_t e = dtab[w & ((1u << LMAX) - 1)];
        const uint32_t len = e >> 8;
        w >>= len;
        bp += len;
        return e & 255u;
    };
    auto four = [&]() -> uint32_t {                                         // 4 symbols, packed
        uint32_t w = window();
        uint32_t d = sym(w);
        d |= sym(w) << 8;
        w = window();
        d |= sym(w) << 16;
        d |= sym(w) << 24;
        return d;
    };

    uint32_t i = a;
    auto sixteen = [&]() -> u32x4 {
        refill();
        u32x4 r;
        r.x = four();
        r.y = four();
        r.z = four();
        r.w = four();
        return r;
    };
    // bytes up to a 16-byte boundary of the output, 16-byte steps up to a 64-byte boundary,
    // then the 64-byte body; the same in reverse at the end
    {
        uint32_t pro = (16u - (uint32_t)((ooff + a) & 15)) & 15u;
        if (pro > b - a) pro = b - a;
        if (pro) refill();
        for (uint32_t k = 0; k < pro; k++) {
            uint32_t w = window();
            o[i++] = (uint8_t)sym(w);
        }
    }
    while (((ooff + i) & 63) != 0 && i + 16 <= b) {
        *(u32x4_a1*)(o + i) = sixteen();
        i += 16;
    }
    // The 64-byte body.  A lane's four 16-byte stores would be four separate memory requests;
    // the four lanes of a quad (= the four sub-streams of one chunk) transpose their pieces in
    // registers instead (two DPP butterfly stages), so that each store instruction writes one
    // member's 64 bytes as ONE request: lane p of the quad stores piece p of member q in round q.
    // The loop runs while any member of the quad has a body left (DPP reads 0 from lanes that
    // are switched off, so the exchange must run with the whole quad enabled).
    {
        const int lane = t & 63;
        const bool odd1 = (t & 1) != 0, odd2 = (t & 2) != 0;
        const uint32_t p = (uint32_t)t & 3u;
        for (;;) {
            const bool work = i + 64 <= b;
            const uint64_t any = __ballot(work);
            if (((any >> (lane & ~3)) & 0xfull) == 0) break;                  // quad-uniform
            uint32_t v[4][4];                                                 // [piece][dword]
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int d = 0; d < 4; d++) v[q][d] = 0;
            if (work) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    // refill(), with the four loads of the piece spread over the period
                    if (have_pend) {
                        park(fpiece, pend);
                        fpiece++;
                    }
                    have_pend = fpiece - (bp >> 9) < 2u;
                    const uint64_t q0 = (piece0 + fpiece) << 2;
                    auto part = [&](int m) {
                        uint64_t qq = q0 + m;
                        qq = qq < last16 ? qq : last16;
                        pend.v[m] = *(const u32x4_a4*)(have_pend ? huf + (qq << 4) : idle);
                    };
                    part(0);
                    v[q][0] = four();
                    part(1);
                    v[q][1] = four();
                    part(2);
                    v[q][2] = four();
                    part(3);
                    v[q][3] = four();
                }
            }
            // 4 x 4 transpose of 16-byte pieces across the quad: (member m, piece k) -> (lane k, slot m)
#pragma unroll
            for (int k = 0; k < 4; k += 2)                                    // stage 1: lanes l ^ 1 swap slots k+1 <-> k
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const uint32_t send = odd1 ? v[k][d] : v[k + 1][d];
                    const uint32_t recv = (uint32_t)__builtin_amdgcn_mov_dpp((int)send, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
                    if (odd1) v[k][d] = recv; else v[k + 1][d] = recv;
                }
#pragma unroll
            for (int k = 0; k < 2; k++)                                       // stage 2: lanes l ^ 2 swap slots k+2 <-> k
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const uint32_t send = odd2 ? v[k][d] : v[k + 2][d];
                    const uint32_t recv = (uint32_t)__builtin_amdgcn_mov_dpp((int)send, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
                    if (odd2) v[k][d] = recv; else v[k + 2][d] = recv;
                }
            const uint64_t mine = work ? (uint64_t)(uintptr_t)(o + i) : 0ull;
            const int mlo = (int)(uint32_t)mine, mhi = (int)(uint32_t)(mine >> 32);
#pragma unroll
            for (int q = 0; q < 4; q++) {                                     // member q's 64 bytes, 16 from each lane
                const int ctrl = q * 0x55;                                    // quad_perm [q,q,q,q]
                uint32_t dlo, dhi;
                if (q == 0) { dlo = (uint32_t)__builtin_amdgcn_mov_dpp(mlo, 0x00, 0xf, 0xf, true); dhi = (uint32_t)__builtin_amdgcn_mov_dpp(mhi, 0x00, 0xf, 0xf, true); }
                else if (q == 1) { dlo = (uint32_t)__builtin_amdgcn_mov_dpp(mlo, 0x55, 0xf, 0xf, true); dhi = (uint32_t)__builtin_amdgcn_mov_dpp(mhi, 0x55, 0xf, 0xf, true); }
                else if (q == 2) { dlo = (uint32_t)__builtin_amdgcn_mov_dpp(mlo, 0xAA, 0xf, 0xf, true); dhi = (uint32_t)__builtin_amdgcn_mov_dpp(mhi, 0xAA, 0xf, 0xf, true); }
                else { dlo = (uint32_t)__builtin_amdgcn_mov_dpp(mlo, 0xFF, 0xf, 0xf, true); dhi = (uint32_t)__builtin_amdgcn_mov_dpp(mhi, 0xFF, 0xf, 0xf, true); }
                (void)ctrl;
                const uint64_t dst = ((uint64_t)dhi << 32) | dlo;
                if (dst) {
                    u32x4 piece = {v[q][0], v[q][1], v[q][2], v[q][3]};
                    *(u32x4_a1*)(uintptr_t)(dst + 16u * p) = piece;
                }
            }
            if (work) i += 64;
        }
    }
    while (i + 16 <= b) {
        *(u32x4_a1*)(o + i) = sixteen();
        i += 16;
    }
    if (i < b) refill();
    for (; i < b; i++) {
        uint32_t w = window();
        o[i] = (uint8_t)sym(w);
    }
}

thread_local std::string g_huf_error;

#include "huf0_write.h"

}  // namespace

extern "C" {

size_t sprintz_mi355x_huf_tmp_bytes(uint64_t nchunks)
{
    const uint64_t nseg = (nchunks + SEG - 1) / SEG;
    // enc tables | meta (u64 per chunk) | record sizes (u32 per chunk) | scan scratch
    return (size_t)(nseg * 1024 + nchunks * 8 + ((nchunks * 4 + 15) & ~(uint64_t)15) + sprintz_mi355x_compact_tmp_bytes(nchunks) + 64);
}

size_t sprintz_mi355x_huf_bound(uint64_t total_stream_bytes, uint64_t nchunks)
{
    return (size_t)(total_stream_bytes + 8 * nchunks + SPRINTZ_MI355X_READ_SLACK);
}

int sprintz_mi355x_huf_compress_batch(const void* d_dense, const uint64_t* d_offsets, const uint32_t* d_sizes, uint64_t nchunks,
                                      void* d_huf, uint64_t* d_huf_offsets, void* d_tables, void* d_tmp, void* hip_stream)
{
    if (!d_dense || !d_offsets || !d_sizes || !d_huf || !d_huf_offsets || !d_tables || !d_tmp) return sprintz::set_error(SPRINTZ_E_INVALID, "Huffman stage: invalid argument (null pointer, alignment or size)");
    hipStream_t st = (hipStream_t)hip_stream;
    if (nchunks == 0) return hipMemsetAsync(d_huf_offsets, 0, 8, st) == hipSuccess ? 0 : sprintz::set_error(SPRINTZ_E_HIP, "Huffman stage: a HIP call or kernel launch failed");
    const uint64_t nseg = (nchunks + SEG - 1) / SEG;
    uint8_t* tmp = (uint8_t*)d_tmp;
    uint32_t* enc_tables = (uint32_t*)tmp;
    uint64_t* meta = (uint64_t*)(tmp + nseg * 1024);
    uint32_t* rec_sizes = (uint32_t*)(tmp + nseg * 1024 + nchunks * 8);
    void* scan_tmp = tmp + nseg * 1024 + nchunks * 8 + ((nchunks * 4 + 15) & ~(uint64_t)15);
    hipLaunchKernelGGL(huf_build_kernel, dim3((unsigned)nseg), dim3(256), 0, st, (const uint8_t*)d_dense, d_offsets, d_sizes, nchunks,
                       enc_tables, (uint8_t*)d_tables);
    hipLaunchKernelGGL(huf_size_kernel, dim3((unsigned)nseg), dim3(256), 0, st, (const uint8_t*)d_dense, d_offsets, d_sizes, nchunks,
                       (const uint32_t*)enc_tables, rec_sizes, meta);
    if (launch_size_scan(rec_sizes, nchunks, 4, d_huf_offsets, scan_tmp, st) != hipSuccess) return sprintz::set_error(SPRINTZ_E_HIP, "Huffman stage: a HIP call or kernel launch failed");
    hipLaunchKernelGGL(huf_encode_kernel, dim3((unsigned)nseg), dim3(256), 0, st, (const uint8_t*)d_dense, d_offsets, d_sizes, nchunks,
                       (const uint32_t*)enc_tables, (const uint64_t*)meta, (uint8_t*)d_huf, (const uint64_t*)d_huf_offsets);
    return hipGetLastError() == hipSuccess ? 0 : sprintz::set_error(SPRINTZ_E_HIP, "Huffman stage: a HIP call or kernel launch failed");
}

// ---- Huff0-format writer (huf0_write.h).  tmp: K1's code tables | nibble tables | segment records | block sizes | meta | scan scratch
size_t sprintz_mi355x_huf0_tmp_bytes(uint64_t nchunks)
{
    const uint64_t nseg = (nchunks + SEG - 1) / SEG;
    return (size_t)(nseg * 1024 + nseg * 128 + nseg * kRecBytes + ((nchunks * 4 + 15) & ~(uint64_t)15) + nchunks * 8 +
                    sprintz_mi355x_compact_tmp_bytes(nchunks) + 64);
}

size_t sprintz_mi355x_huf0_bound(uint64_t total_stream_bytes, uint64_t nchunks)
{
    (void)nchunks;                                        // a block is never larger than its chunk
    return (size_t)(total_stream_bytes + SPRINTZ_MI355X_READ_SLACK);
}

int sprintz_mi355x_huf0_compress_batch(const void* d_dense, const uint64_t* d_offsets, const uint32_t* d_sizes, uint64_t nchunks,
                                       void* d_blocks, uint64_t* d_block_offsets, void* d_tmp, void* hip_stream)
{
    if (!d_dense || !d_offsets || !d_sizes || !d_blocks || !d_block_offsets || !d_tmp) return sprintz::set_error(SPRINTZ_E_INVALID, "Huffman stage: invalid argument (null pointer, alignment or size)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return sprintz::set_error(SPRINTZ_E_NO_DEVICE, "Huffman stage: no usable HIP device (there is no CPU fallback)");
    hipStream_t st = (hipStream_t)hip_stream;
    if (nchunks == 0) return hipMemsetAsync(d_block_offsets, 0, 8, st) == hipSuccess ? 0 : sprintz::set_error(SPRINTZ_E_HIP, "Huffman stage: a HIP call or kernel launch failed");
    const uint64_t nseg = (nchunks + SEG - 1) / SEG;
    uint8_t* tmp = (uint8_t*)d_tmp;
    uint32_t* enc_tables = (uint32_t*)tmp;
    uint8_t* nib = tmp + nseg * 1024;
    uint8_t* recs = nib + nseg * 128;
    uint32_t* bsizes = (uint32_t*)(recs + nseg * kRecBytes);
    uint64_t* meta = (uint64_t*)((uint8_t*)bsizes + ((nchunks * 4 + 15) & ~(uint64_t)15));
    void* scan_tmp = (uint8_t*)meta + nchunks * 8;
    hipLaunchKernelGGL(huf_build_kernel, dim3((unsigned)nseg), dim3(256), 0, st, (const uint8_t*)d_dense, d_offsets, d_sizes, nchunks,
                       enc_tables, nib);
    hipLaunchKernelGGL(huf0_table_kernel, dim3((unsigned)nseg), dim3(64), 0, st, (const uint8_t*)nib, recs);
    hipLaunchKernelGGL(huf0_size_kernel, dim3((unsigned)nseg), dim3(256), nseg >= kSizePassPadFrom ? kSizePassPad : 0, st, (const uint8_t*)d_dense, d_offsets, d_sizes, nchunks,
                       (const uint8_t*)recs, bsizes, meta);
    if (launch_size_scan(bsizes, nchunks, 1, d_block_offsets, scan_tmp, st) != hipSuccess) return sprintz::set_error(SPRINTZ_E_HIP, "Huffman stage: a HIP call or kernel launch failed");
    hipLaunchKernelGGL(huf0_encode_kernel, dim3((unsigned)nseg), dim3(256), 0, st, (const uint8_t*)d_dense, d_offsets, d_sizes, nchunks,
                       (const uint8_t*)recs, (const uint64_t*)meta, (uint8_t*)d_blocks, (const uint64_t*)d_block_offsets);
    return hipGetLastError() == hipSuccess ? 0 : sprintz::set_error(SPRINTZ_E_HIP, "Huffman stage: a HIP call or kernel launch failed");
}

int sprintz_mi355x_huf_decompress_batch(const void* d_huf, const uint64_t* d_huf_offsets, const void* d_tables, uint64_t nchunks,
                                        uint32_t align, void* d_dense, uint64_t dense_capacity, uint64_t* d_offsets,
                                        uint32_t* d_sizes, int64_t* d_rets, void* d_tmp, void* hip_stream)
{
    if (!d_huf || !d_huf_offsets || !d_tables || !d_dense || !d_offsets || !d_sizes || !d_tmp) return sprintz::set_error(SPRINTZ_E_INVALID, "Huffman stage: invalid argument (null pointer, alignment or size)");
    if (align == 0 || align > 16 || (align & (align - 1))) return sprintz::set_error(SPRINTZ_E_INVALID, "Huffman stage: invalid argument (null pointer, alignment or size)");
    hipStream_t st = (hipStream_t)hip_stream;
    if (nchunks == 0) return hipMemsetAsync(d_offsets, 0, 8, st) == hipSuccess ? 0 : sprintz::set_error(SPRINTZ_E_HIP, "Huffman stage: a HIP call or kernel launch failed");
    const uint64_t nseg = (nchunks + SEG - 1) / SEG;
    hipLaunchKernelGGL(huf_rawsize_kernel, dim3((unsigned)((nchunks + 255) / 256)), dim3(256), 0, st, (const uint8_t*)d_huf, d_huf_offsets,
                       nchunks, d_sizes);
    if (launch_size_scan(d_sizes, nchunks, align, d_offsets, d_tmp, st) != hipSuccess) return sprintz::set_error(SPRINTZ_E_HIP, "Huffman stage: a HIP call or kernel launch failed");
    hipLaunchKernelGGL(huf_decode_kernel, dim3((unsigned)nseg), dim3(256), 0, st, (const uint8_t*)d_huf, d_huf_offsets,
                       (const uint8_t*)d_tables, nchunks, (uint8_t*)d_dense, (const uint64_t*)d_offsets, dense_capacity, d_rets);
    return hipGetLastError() == hipSuccess ? 0 : sprintz::set_error(SPRINTZ_E_HIP, "Huffman stage: a HIP call or kernel launch failed");
}

#ifdef HUF_BUILD_TIMING
int sprintz_mi355x_dbg_build_stamps(uint64_t* out16) { return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_build_ts), 16 * sizeof(uint64_t)); }
#endif

}  // extern "C"

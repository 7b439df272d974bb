// online.hip -- the reference's 2020 "online" coders for 1-D uint16 streams (cpp/Compress/online.hpp:395-445,
// online.cpp) on gfx950:
//   dynamic_delta_pack_u16 / _altloss / dynamic_delta_unpack_u16     online.cpp:48-311
//   zigzag_pack_u16 / zigzag_unpack_u16                               online.cpp:314-351
//   sprintzpack_pack_u16 / _zigzag / sprintzpack_unpack_u16 / _zigzag online.cpp:355-703
// Container formats and every quirk are restated, with citations, in oracle/online_oracle.c (pinned against the
// compiled reference).  One call codes ONE stream of any length; the reference walks it serially, here:
//   * encoders are block-parallel: a block's predictor choice / bit width is a pure function of the input (both of the
//     reference's predictors are trained on the true values, so neither depends on earlier choices);
//   * sprintzpack's byte offsets are an exclusive scan of the per-block widths: inside a tile of 1 024 blocks, the tiles' sums scanned by one workgroup;
//   * the dynamic-delta DECODER is a scan too: a block maps the running state (x, d) = (last value, last difference)
//     affinely -- delta: (x + A, e7); double delta: (x + 8 d + C, d + A) with A = sum e, C = sum (8 - i) e_i -- and such
//     maps compose associatively as (m, a, tx, td): x' = x + m d + tx, d' = a d + td, everything modulo 2^16.
//     Three launches: tile summaries (1 024 blocks = 8 192 samples a workgroup), one workgroup scanning the tiles,
//     and the decode proper with every thread's incoming state.
#include "../../include/sprintz_mi355x.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "launch.h"

using namespace sprintz;

namespace {

constexpr int kT = 256;                                   // threads per workgroup
constexpr int kBPT = 4;                                   // blocks of 8 samples a thread takes (consecutive: 64 bytes of input)
constexpr int kTile = kT * kBPT;                          // blocks per workgroup = per scan tile (8 192 samples)

typedef uint16_t __attribute__((aligned(1), may_alias)) u16_a1;
typedef uint32_t __attribute__((aligned(1), may_alias)) u32_a1;
typedef uint32_t v4 __attribute__((ext_vector_type(4)));
typedef v4 __attribute__((aligned(2), may_alias)) v4a2;  // 8 samples at any SAMPLE boundary: one 16-byte request (the containers' bodies start 4 or 6 bytes in)
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t zz16(uint32_t x)      // zigzag_encode_16b of the int16 in the low half (bitpack.h:312)
{
    const int v = (int)(int16_t)x;
    return ((uint32_t)(v << 1) ^ (uint32_t)(v >> 15)) & 0xffffu;
}
__device__ __forceinline__ uint32_t unzz16(uint32_t z) { return ((z >> 1) ^ (0u - (z & 1u))) & 0xffffu; }   // bitpack.h:315
// the same on both halves of a dword (v_pk_* instructions)
__device__ __forceinline__ uint32_t zz16x2(uint32_t w)
{
    const s16x2 v = __builtin_bit_cast(s16x2, w);
    return __builtin_bit_cast(uint32_t, (s16x2)((v << 1) ^ (v >> 15)));
}
__device__ __forceinline__ uint32_t unzz16x2(uint32_t w)
{
    const u16x2 z = __builtin_bit_cast(u16x2, w);
    const u16x2 one = {1, 1}, zero = {0, 0};
    return __builtin_bit_cast(uint32_t, (u16x2)((z >> 1) ^ (zero - (z & one))));
}

__device__ __forceinline__ uint32_t ld16(const uint8_t* p) { return *(const u16_a1*)p; }
__device__ __forceinline__ void st16(uint8_t* p, uint32_t v) { *(u16_a1*)p = (uint16_t)v; }
__device__ __forceinline__ uint32_t half_of(const v4& q, int i) { const uint32_t d = (i >> 1) == 0 ? q.x : (i >> 1) == 1 ? q.y : (i >> 1) == 2 ? q.z : q.w; return (i & 1) ? d >> 16 : d & 0xffffu; }

// sum over the workgroup (every thread gets it); sh: kT / 64 words
__device__ __forceinline__ uint32_t workgroup_sum(uint32_t v, uint32_t* sh)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    uint32_t s = 0;
#pragma unroll
    for (int w = 0; w < kT / 64; w++) s += sh[w];
    __syncthreads();
    return s;
}
// exclusive prefix sum over the workgroup; sh: kT / 64 words
__device__ __forceinline__ uint32_t workgroup_excl(uint32_t v, uint32_t* sh)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)inc, d); if (lane >= d) inc += u; }
    if (lane == 63) sh[w] = inc;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int k = 0; k < kT / 64; k++) base += k < w ? sh[k] : 0u;
    __syncthreads();
    return base + inc - v;
}

// what an unpack launch reports: the stream's header (the four bytes in front of its body) must agree with the caller's length -- checked by the
// thread that writes the return value (round 5: this was a launch of its own behind the decode, and a memset of the return word in front of it)
__device__ __forceinline__ int64_t checked_len(const uint8_t* body, uint32_t len)
{
    const uint32_t have = (uint32_t)body[-4] | ((uint32_t)body[-3] << 8) | ((uint32_t)body[-2] << 16) | ((uint32_t)body[-1] << 24);
    return have == len ? (int64_t)len : (int64_t)SPRINTZ_E_CORRUPT;
}

// ---------------------------------------------------------------- zigzag
// src / dst: the first SAMPLE on both sides (one of them 4 bytes into its buffer); 8 samples = one 16-byte request a thread
__global__ void __launch_bounds__(kT) zigzag_kernel(const uint8_t* src, uint8_t* dst, uint32_t len, int decode, int64_t* ret)
{
    const uint64_t i0 = ((uint64_t)blockIdx.x * kT + threadIdx.x) * 8;
    if (i0 + 8 <= len) {
        v4 q = *(const v4a2*)(src + 2 * i0);
        if (decode) { q.x = unzz16x2(q.x); q.y = unzz16x2(q.y); q.z = unzz16x2(q.z); q.w = unzz16x2(q.w); }
        else { q.x = zz16x2(q.x); q.y = zz16x2(q.y); q.z = zz16x2(q.z); q.w = zz16x2(q.w); }
        *(v4a2*)(dst + 2 * i0) = q;
    } else {
        for (uint64_t i1 = i0; i1 < len; i1++) st16(dst + 2 * i1, decode ? unzz16(ld16(src + 2 * i1)) : zz16(ld16(src + 2 * i1)));
    }
    if (i0 == 0 && ret) *ret = decode ? checked_len(src, len) : 2 + (int64_t)len;
}

// ---------------------------------------------------------------- dynamic delta, encoder
// thread = block b of 8 samples (elements 1 + 8 b ... 8 + 8 b): one 16-byte request for them (2 bytes off the 16-byte grid), one
// 4-byte request for the two samples in front, one 16-byte store.  x: the samples; out: the container's sample area
__global__ void __launch_bounds__(kT) dyndelta_encode_kernel(const uint16_t* x, uint32_t len, uint8_t* out, uint8_t* choices,
                                                             uint32_t choice_bytes, int alt, int64_t* ret)
{
    const uint32_t n = len - 1, nblocks = n / 8;           // len >= 2 here
    const uint32_t b = blockIdx.x * kT + threadIdx.x;
    int choice = 0;
    if (b < nblocks) {
        const uint32_t at0 = 1 + 8 * b;
        const v4 q = *(const v4a2*)(x + at0);
        uint32_t p1, p2;
        if (b) { const uint32_t w = *(const u32_a1*)(x + at0 - 2); p2 = w & 0xffffu; p1 = w >> 16; }
        else { p1 = x[0]; p2 = x[0]; }
        uint32_t z0[8], z1[8], m0 = 0, m1 = 0, s0 = 0, s1 = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t v = half_of(q, i);
            z0[i] = zz16(v - p1);                                  // delta                       online.hpp: DeltaPredictor_u16
            z1[i] = zz16(v - (2 * p1 - p2));                       // double delta                online.hpp: DoubleDeltaPredictor_u16
            p2 = p1;
            p1 = v;
            m0 = z0[i] > m0 ? z0[i] : m0;
            m1 = z1[i] > m1 ? z1[i] : m1;
            // SumLogAbs as the reference's compiled code computes it (online.cpp:36-43): (uint8_t)(16 - clz32(v)), lzcnt(0) = 32
            s0 += (uint32_t)(16 - __clz((int)z0[i])) & 0xffu;
            s1 += (uint32_t)(16 - __clz((int)z1[i])) & 0xffu;
        }
        choice = alt ? (m0 <= m1 ? 0 : 1) : (s0 <= s1 ? 0 : 1);    // loss0 <= loss1 keeps delta (:119)
        v4 o;
        o.x = (choice ? z1[0] : z0[0]) | ((choice ? z1[1] : z0[1]) << 16);
        o.y = (choice ? z1[2] : z0[2]) | ((choice ? z1[3] : z0[3]) << 16);
        o.z = (choice ? z1[4] : z0[4]) | ((choice ? z1[5] : z0[5]) << 16);
        o.w = (choice ? z1[6] : z0[6]) | ((choice ? z1[7] : z0[7]) << 16);
        *(v4a2*)(out + 2 * (uint64_t)at0) = o;
    }
    // one choice bit per block, LSB first: a wavefront's 64 bits leave as 8 bytes
    const uint64_t bits = __ballot(choice != 0);
    const uint32_t lane = threadIdx.x & 63u, byte0 = (b - lane) / 8;
    if (lane < 8 && byte0 + lane < choice_bytes) choices[byte0 + lane] = (uint8_t)(bits >> (8 * lane));
    if (b == 0) {
        st16(out, x[0]);                                           // element 0 verbatim (:57)
        for (uint32_t at = 1 + 8 * nblocks; at < len; at++) st16(out + 2 * (uint64_t)at, (uint32_t)x[at] - (uint32_t)x[at - 1]);   // tail: delta, no zigzag (:149-155)
        if (ret) *ret = 2 + (int64_t)len + (choice_bytes + 1) / 2;
    }
}

// ---------------------------------------------------------------- dynamic delta, decoder (scan)
struct Aff { uint32_t m, a, tx, td; };                    // x' = x + m d + tx ; d' = a d + td   (mod 2^16)
__device__ __forceinline__ Aff compose(const Aff& f, const Aff& g)   // f first, then g
{
    Aff r;
    r.m = (f.m + g.m * f.a) & 0xffffu;
    r.a = f.a * g.a;
    r.tx = (f.tx + g.m * f.td + g.tx) & 0xffffu;
    r.td = (g.a * f.td + g.td) & 0xffffu;
    return r;
}
__device__ __forceinline__ uint64_t pack_aff(const Aff& f) { return (uint64_t)f.m | ((uint64_t)f.a << 16) | ((uint64_t)f.tx << 32) | ((uint64_t)f.td << 48); }
__device__ __forceinline__ Aff unpack_aff(uint64_t v) { return Aff{(uint32_t)v & 0xffffu, (uint32_t)(v >> 16) & 1u, (uint32_t)(v >> 32) & 0xffffu, (uint32_t)(v >> 48)}; }
constexpr uint64_t kIdentity = (uint64_t)1 << 16;         // m = 0, a = 1, tx = td = 0

// block b's map from its 8 zigzagged errors (one 16-byte request); e[] receives the errors
__device__ __forceinline__ Aff block_map(const uint8_t* in, const uint8_t* choices, uint32_t b, uint32_t (&e)[8], int& choice)
{
    choice = (choices[b / 8] >> (b % 8)) & 1;
    v4 q = *(const v4a2*)(in + 2 * (uint64_t)(1 + 8 * b));
    q.x = unzz16x2(q.x); q.y = unzz16x2(q.y); q.z = unzz16x2(q.z); q.w = unzz16x2(q.w);
    uint32_t A = 0, C = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        e[i] = half_of(q, i);
        A += e[i];
        C += (uint32_t)(8 - i) * e[i];
    }
    return choice ? Aff{8u, 1u, C & 0xffffu, A & 0xffffu} : Aff{0u, 0u, A & 0xffffu, e[7]};
}

__device__ __forceinline__ uint64_t shfl_up64(uint64_t v, int d)
{
    return ((uint64_t)(uint32_t)__shfl_up((int)(uint32_t)(v >> 32), d) << 32) | (uint32_t)__shfl_up((int)(uint32_t)v, d);
}
// scan of one map per thread over the workgroup (composition is not commutative: left operand = earlier).  Returns the composition of
// the threads BEFORE this one; total = all of them.  sh: kT / 64 words of 64 bits
template <int T = kT>
__device__ __forceinline__ Aff workgroup_scan_excl(const Aff& mine, Aff& total, uint64_t* sh)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint64_t inc = pack_aff(mine);
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t prev = shfl_up64(inc, d);
        if (lane >= d) inc = pack_aff(compose(unpack_aff(prev), unpack_aff(inc)));
    }
    uint64_t before = shfl_up64(inc, 1);
    if (lane == 0) before = kIdentity;
    if (lane == 63) sh[w] = inc;
    __syncthreads();
    Aff base = unpack_aff(kIdentity), all = unpack_aff(kIdentity);
#pragma unroll
    for (int k = 0; k < T / 64; k++) {
        const Aff wk = unpack_aff(sh[k]);
        if (k < w) base = compose(base, wk);
        all = compose(all, wk);
    }
    __syncthreads();
    total = all;
    return compose(base, unpack_aff(before));
}

constexpr int kDdBPT = 4;                                 // blocks a thread of the dynamic-delta decoder takes (8 measured: 0.222 against 0.135 ms -- 64 more registers of errors a lane)
constexpr int kDdTile = kT * kDdBPT;
// Three launches: tile summaries (a tile = kDdTile blocks, a thread kDdBPT consecutive ones, their maps composed in order), one workgroup
// scanning the tiles, and the decode proper with every thread's incoming state.  (Round 5 also built the ONE-launch form -- the state in
// front of a tile from a chained scan over tiles, decoupled look-back as in compact_tail.h -- and measured it slower: 0.166 against
// 0.129 ms for 64 Mi samples, 0.131 with the look-back cut out.  With 8 192 tiles resident at once the first cohort's look-backs are a
// chain of ~24 dependent device-scope round trips; a 60 us kernel has nothing to hide that behind.  Same finding for sprintzpack below.)
__global__ void __launch_bounds__(kT) dyndelta_tile_kernel(const uint8_t* in, const uint8_t* choices, uint32_t nblocks, uint64_t* tiles)
{
    __shared__ uint64_t sh[kT / 64];
    const uint32_t b0 = (blockIdx.x * kT + threadIdx.x) * kDdBPT;
    Aff f = unpack_aff(kIdentity);
#pragma unroll
    for (int j = 0; j < kDdBPT; j++)
        if (b0 + j < nblocks) { uint32_t e[8]; int c; f = compose(f, block_map(in, choices, b0 + j, e, c)); }
    Aff total;
    (void)workgroup_scan_excl(f, total, sh);
    if (threadIdx.x == 0) tiles[blockIdx.x] = pack_aff(total);
}

// one workgroup: tiles[i] <- composition of tiles[0 .. i-1] (exclusive).  The tiles pass through LDS in slabs of kT * kSlab (read
// and written coalesced; a thread's kSlab consecutive ones are then LDS reads -- as dependent global loads they were 20 us of the call)
// (round 5: 1 024 threads x 4 tiles instead of 256 x 16 -- one workgroup's serial phases were 16.5 us of a 130 us unpack)
constexpr int kSlab = 4, kDdScanT = 1024;
__global__ void __launch_bounds__(kDdScanT) dyndelta_tilescan_kernel(uint64_t* tiles, uint32_t ntiles)
{
    constexpr int kT = kDdScanT;                       // (shadows the file's 256 inside this kernel)
    __shared__ uint64_t slab[kT * kSlab];
    __shared__ uint64_t sh[kT / 64];
    Aff carry = unpack_aff(kIdentity);
    for (uint32_t base = 0; base < ntiles; base += kT * kSlab) {
        for (int k = 0; k < kSlab; k++) {
            const uint32_t i = base + k * kT + threadIdx.x;
            slab[k * kT + threadIdx.x] = i < ntiles ? tiles[i] : kIdentity;
        }
        __syncthreads();
        Aff mine = unpack_aff(kIdentity);
        for (int k = 0; k < kSlab; k++) mine = compose(mine, unpack_aff(slab[threadIdx.x * kSlab + k]));
        Aff total;
        Aff run = compose(carry, workgroup_scan_excl<kT>(mine, total, sh));
        for (int k = 0; k < kSlab; k++) {
            const Aff t = unpack_aff(slab[threadIdx.x * kSlab + k]);
            slab[threadIdx.x * kSlab + k] = pack_aff(run);
            run = compose(run, t);
        }
        carry = compose(carry, total);
        __syncthreads();
        for (int k = 0; k < kSlab; k++) {
            const uint32_t i = base + k * kT + threadIdx.x;
            if (i < ntiles) tiles[i] = slab[k * kT + threadIdx.x];
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(kT) dyndelta_decode_kernel(const uint8_t* in, const uint8_t* choices, uint32_t len, uint32_t nblocks,
                                                             const uint64_t* tiles, uint16_t* out, int64_t* ret)
{
    __shared__ uint64_t sh[kT / 64];
    const uint32_t b0 = (blockIdx.x * kT + threadIdx.x) * kDdBPT;
    uint32_t e[kDdBPT][8];
    int choice[kDdBPT];
    Aff f = unpack_aff(kIdentity);
#pragma unroll
    for (int j = 0; j < kDdBPT; j++) {
        choice[j] = 0;
        if (b0 + j < nblocks) f = compose(f, block_map(in, choices, b0 + j, e[j], choice[j]));
    }
    Aff total;
    const Aff before_wg = workgroup_scan_excl(f, total, sh);
    // state before this thread's first block: (x0, 0) through the tiles before, then the threads before
    const Aff before = compose(unpack_aff(tiles[blockIdx.x]), before_wg);
    const uint32_t x0 = ld16(in);
    uint32_t x = (x0 + before.tx) & 0xffffu, d = before.td;       // d starts at 0 (online.hpp: _prev_diff = 0)
#pragma unroll
    for (int j = 0; j < kDdBPT; j++) {
        if (b0 + j < nblocks) {
            uint32_t v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                d = choice[j] ? (d + e[j][i]) & 0xffffu : e[j][i];
                x = (x + d) & 0xffffu;
                v[i] = x;
            }
            v4 o;
            o.x = v[0] | (v[1] << 16); o.y = v[2] | (v[3] << 16); o.z = v[4] | (v[5] << 16); o.w = v[6] | (v[7] << 16);
            *(v4a2*)(out + 1 + 8 * (uint64_t)(b0 + j)) = o;
            if (b0 + j == nblocks - 1) {                          // the owner of the last block walks the < 8 trailing delta errors (:245-251)
                for (uint32_t at = 1 + 8 * nblocks; at < len; at++) {
                    x = (x + ld16(in + 2 * (uint64_t)at)) & 0xffffu;
                    out[at] = (uint16_t)x;
                }
            }
        }
    }
    if (b0 == 0) {
        out[0] = (uint16_t)x0;
        if (nblocks == 0) {                                       // fewer than 9 elements: only the trailing deltas
            uint32_t y = x0;
            for (uint32_t at = 1; at < len; at++) { y = (y + ld16(in + 2 * (uint64_t)at)) & 0xffffu; out[at] = (uint16_t)y; }
        }
        if (ret) *ret = checked_len(in, len);
    }
}

// ---------------------------------------------------------------- dynamic delta, decoder in ONE pass (round 6)
// transforms.hip's chained scan with this coder's affine maps as the monoid: a persistent workgroup of 8 waves a CU takes TILES of
// 8 192 blocks (128 KB of errors, kept packed in registers: 16 blocks a lane as 4 loads of 4 consecutive blocks, handed over in LDS so that
// loads and stores are coalesced), folds them, publishes the tile's map, gets the map of everything BEFORE the tile from a look-back by all 8
// waves at once (a lane a predecessor, a DPP scan from the nearest published state on, the waves' results folded in LDS), publishes the state,
// decodes and stores.  A tile's word is its packed map (64 bits, 15 of them spare) with a tag in bits 24 - 25 -- 1: the tile's own map, 2: the
// map up to and including it -- one relaxed agent-scope atomic word, no fence, no flag beside it.  Round 5's one-launch form (one wave looking
// back over tiles of 1 024 blocks, 8 192 of them resident) measured 0.166 against 0.129 ms; this one 0.114 -> see DESIGN 4.10.
constexpr int kDcWaves = 8, kDcT = 64 * kDcWaves, kDcJ = 4, kDcK = 4;
constexpr uint32_t kDcWaveBlocks = kDcJ * 64 * kDcK, kDcTileBlocks = kDcWaves * kDcWaveBlocks;     // 1 024, 8 192
constexpr uint64_t kDcTagMap = 1ull << 24, kDcTagState = 2ull << 24, kDcTagMask = 3ull << 24;

__device__ __forceinline__ void dc_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ uint32_t dc_swz(uint32_t p) { return (p & ~3u) | ((p + (p >> 4)) & 3u); }
// the lane's neighbour's map through a DPP move; a lane without a source reads the identity
template <int CTRL, int RM> __device__ __forceinline__ Aff dpp_aff(const Aff& f)
{
    Aff r;
    r.m = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)f.m, CTRL, RM, 0xf, false);
    r.a = (uint32_t)__builtin_amdgcn_update_dpp(1, (int)f.a, CTRL, RM, 0xf, false);
    r.tx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)f.tx, CTRL, RM, 0xf, false);
    r.td = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)f.td, CTRL, RM, 0xf, false);
    return r;
}
// inclusive scan of N maps a lane over the wave, side by side (lane l: the composition of lanes 0 .. l)
template <int N> __device__ __forceinline__ void wave_scan_aff(Aff (&v)[N])
{
#define DC_STEP(CTRL, RM) _Pragma("unroll") for (int j = 0; j < N; j++) v[j] = compose(dpp_aff<CTRL, RM>(v[j]), v[j]);
    DC_STEP(0x111, 0xf) DC_STEP(0x112, 0xf) DC_STEP(0x114, 0xf) DC_STEP(0x118, 0xf) DC_STEP(0x142, 0xa) DC_STEP(0x143, 0xc)
#undef DC_STEP
}
__device__ __forceinline__ Aff lane63_aff(const Aff& f)
{
    return Aff{(uint32_t)__builtin_amdgcn_readlane((int)f.m, 63), (uint32_t)__builtin_amdgcn_readlane((int)f.a, 63),
               (uint32_t)__builtin_amdgcn_readlane((int)f.tx, 63), (uint32_t)__builtin_amdgcn_readlane((int)f.td, 63)};
}
// a block's map from its 8 errors (packed pairs, zigzag undone)
__device__ __forceinline__ Aff map_of(const v4& e, int choice)
{
    uint32_t A = 0, C = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t v = half_of(e, i);
        A += v;
        C += (uint32_t)(8 - i) * v;
    }
    return choice ? Aff{8u, 1u, C & 0xffffu, A & 0xffffu} : Aff{0u, 0u, A & 0xffffu, e.w >> 16};
}

#ifndef DC_WAVES_PER_EU
#define DC_WAVES_PER_EU 2
#endif
#ifndef DC_WGS_PER_CU
#define DC_WGS_PER_CU 1
#endif
__global__ void __launch_bounds__(kDcT) __attribute__((amdgpu_waves_per_eu(DC_WAVES_PER_EU))) dyndelta_chain_kernel(const uint8_t* in, const uint8_t* choices, uint32_t len, uint32_t nblocks, uint32_t ntiles,
                                                              uint32_t* ticket, uint64_t* words, uint16_t* out, int64_t* ret)
{
    constexpr int J = kDcJ, K = kDcK;
    __shared__ v4 xbuf[kDcWaves][64 * K];
    __shared__ uint64_t sm[kDcWaves];
    __shared__ uint32_t sminc[kDcWaves], s_ticket;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    v4* const xb = xbuf[w];
    const uint32_t x0 = ld16(in);
    if (tid == 0) s_ticket = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    uint32_t tile = s_ticket;
    __syncthreads();
    while (tile < ntiles) {
        if (tid == 0) s_ticket = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (read behind the first barrier below)
        const uint32_t wb0 = tile * kDcTileBlocks + w * kDcWaveBlocks;         // the wave's first block
        // ---- the wave's 4 loads of 256 blocks, coalesced (lane l: blocks m * 64 + l of the load), and the lane's 16 choice bits
        v4 e[J][K];
        uint32_t ch = 0;
#pragma unroll
        for (int j = 0; j < J; j++) {
#pragma unroll
            for (int m = 0; m < K; m++) {
                const uint32_t b = wb0 + (uint32_t)j * 256u + (uint32_t)m * 64u + lane;
                v4 q = {0u, 0u, 0u, 0u};
                if (b < nblocks) q = *(const v4a2*)(in + 2 * (uint64_t)(1 + 8 * (uint64_t)b));
                e[j][m] = q;
            }
            const uint32_t bl = wb0 + (uint32_t)j * 256u + 4u * lane;             // the 4 consecutive blocks this lane folds: half a choice byte
            const uint32_t cbyte = bl < nblocks ? choices[bl >> 3] : 0u;
            ch |= ((cbyte >> (bl & 4u)) & 15u) << (4 * j);
        }
#pragma unroll
        for (int j = 0; j < J; j++) {
#pragma unroll
            for (int m = 0; m < K; m++) xb[dc_swz((uint32_t)m * 64u + lane)] = e[j][m];
            dc_wave_sync();
#pragma unroll
            for (int k = 0; k < K; k++) {
                v4 q = xb[dc_swz(4u * lane + (uint32_t)k)];
                q.x = unzz16x2(q.x); q.y = unzz16x2(q.y); q.z = unzz16x2(q.z); q.w = unzz16x2(q.w);
                e[j][k] = q;
            }
            dc_wave_sync();
        }
        // ---- the lanes' maps of every load, the 4 scans side by side; q[j]: the wave's blocks in front of this lane's of load j
        Aff q[J], W;
        {
            Aff v[J];
#pragma unroll
            for (int j = 0; j < J; j++) {
                Aff f = unpack_aff(kIdentity);
#pragma unroll
                for (int k = 0; k < K; k++) f = compose(f, map_of(e[j][k], (int)((ch >> (4 * j + k)) & 1u)));
                v[j] = f;
            }
            wave_scan_aff<J>(v);
            Aff before = unpack_aff(kIdentity);
#pragma unroll
            for (int j = 0; j < J; j++) {
                q[j] = compose(before, dpp_aff<0x138, 0xf>(v[j]));                // wave_shr:1: the lanes before mine (lane 0: the identity)
                before = compose(before, lane63_aff(v[j]));
            }
            W = before;
        }
        if (lane == 0) sm[w] = pack_aff(W);
        __syncthreads();
        const uint32_t next = s_ticket;
        Aff B = unpack_aff(kIdentity), T = unpack_aff(kIdentity);                 // the waves before mine; the whole tile
#pragma unroll
        for (int k = 0; k < kDcWaves; k++) {
            const Aff wk = unpack_aff(sm[k]);
            if ((uint32_t)k < w) B = compose(B, wk);
            T = compose(T, wk);
        }
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&words[tile], pack_aff(T) | (tile == 0 ? kDcTagState : kDcTagMap), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

        // ---- the map of everything before the tile: lane l of wave w <- the tile at distance 64 w + 63 - l (older tiles in lower lanes)
        Aff A = unpack_aff(kIdentity);
        if (tile != 0) {
            int64_t base = (int64_t)tile - 1;
            for (;;) {
                const int64_t p = base - (int64_t)(64u * w + 63u - lane);
                uint64_t word = kIdentity | kDcTagState;                          // before tile 0: the identity, a state
                if (p >= 0) {
                    word = __hip_atomic_load(&words[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    while ((word & kDcTagMask) == 0ull) {                         // its holder is running (tickets): this ends
                        __builtin_amdgcn_s_sleep(1);
                        word = __hip_atomic_load(&words[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                const uint64_t st_lanes = __ballot((word & kDcTagMask) == kDcTagState);
                const uint32_t nearest = st_lanes ? 63u - (uint32_t)__builtin_clzll(st_lanes) : 0u;      // the nearest state of my window: the highest lane that has one
                Aff v[1] = {lane >= nearest || st_lanes == 0ull ? unpack_aff(word) : unpack_aff(kIdentity)};
                wave_scan_aff<1>(v);
                const Aff tot = lane63_aff(v[0]);
                if (lane == 0) { sm[w] = pack_aff(tot); sminc[w] = st_lanes != 0ull ? 1u : 0u; }
                __syncthreads();
                Aff P = unpack_aff(kIdentity);
                bool found = false;
                for (int k = 0; k < kDcWaves && !found; k++) {                    // wave 0's window is the nearest
                    P = compose(unpack_aff(sm[k]), P);
                    found = sminc[k] != 0u;
                }
                A = compose(P, A);
                __syncthreads();
                if (found) break;
                base -= 64 * kDcWaves;
            }
            if (tid == 0) __hip_atomic_store(&words[tile], pack_aff(compose(A, T)) | kDcTagState, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }

        // ---- decode: the state in front of the lane's blocks of load j; the samples change hands in LDS again and leave coalesced
        const Aff AB = compose(A, B);
#pragma unroll
        for (int j = 0; j < J; j++) {
            const Aff f = compose(AB, q[j]);
            uint32_t x = (x0 + f.tx) & 0xffffu, d = f.td;                         // d starts at 0 (online.hpp: _prev_diff = 0)
            const uint32_t bl = wb0 + (uint32_t)j * 256u + 4u * lane;
#pragma unroll
            for (int k = 0; k < K; k++) {
                const int choice = (int)((ch >> (4 * j + k)) & 1u);
                uint32_t v[8];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const uint32_t ei = half_of(e[j][k], i);
                    d = choice ? (d + ei) & 0xffffu : ei;
                    x = (x + d) & 0xffffu;
                    v[i] = x;
                }
                v4 o;
                o.x = v[0] | (v[1] << 16); o.y = v[2] | (v[3] << 16); o.z = v[4] | (v[5] << 16); o.w = v[6] | (v[7] << 16);
                xb[dc_swz(4u * lane + (uint32_t)k)] = o;
                if (bl + (uint32_t)k == nblocks - 1u) {                           // the owner of the last block walks the < 8 trailing delta errors (:245-251)
                    uint32_t y = x;
                    for (uint32_t at = 1 + 8 * nblocks; at < len; at++) {
                        y = (y + ld16(in + 2 * (uint64_t)at)) & 0xffffu;
                        out[at] = (uint16_t)y;
                    }
                }
            }
            dc_wave_sync();
#pragma unroll
            for (int m = 0; m < K; m++) {
                const uint32_t b = wb0 + (uint32_t)j * 256u + (uint32_t)m * 64u + lane;
                if (b < nblocks) *(v4a2*)(out + 1 + 8 * (uint64_t)b) = xb[dc_swz((uint32_t)m * 64u + lane)];
            }
            dc_wave_sync();
        }
        if (tile == 0 && tid == 0) {
            out[0] = (uint16_t)x0;
            if (ret) *ret = checked_len(in, len);
        }
        tile = next;
    }
}

// ---------------------------------------------------------------- sprintzpack
// A workgroup takes a TILE of kTile blocks (a thread kBPT consecutive ones).  Three launches either way: the tiles' payload sizes,
// one workgroup scanning them, and the pack / unpack proper -- which recomputes its blocks' widths (a re-read of the samples / of the
// header nibbles costs less than round 4's widths array and per-block offsets array written and read back: 12 bytes a block against 16
// of data).  The one-launch form (a chained scan over tiles, as tried for the dynamic-delta decoder above) measured 0.152 against 0.097 ms.
__device__ __forceinline__ uint32_t block_width(const v4& q, int zig)
{
    const uint32_t all = zig ? (zz16x2(q.x) | zz16x2(q.y) | zz16x2(q.z) | zz16x2(q.w)) : (q.x | q.y | q.z | q.w);
    const uint32_t both = (all | (all >> 16)) & 0xffffu;
    uint32_t nb = both == 0 ? 0u : 32u - (uint32_t)__clz((int)both);
    nb += nb == 15u;                                               // bitpack.h:286
    return nb;
}

// payload bytes of every tile + the header nibbles (two blocks a byte: nb - (nb == 16), online.cpp:405-412; bytes past the last block: 0)
__global__ void __launch_bounds__(kT) pack_tile_kernel(const uint16_t* x, uint32_t nblocks, int zig, uint8_t* hdr, uint32_t hdr_bytes, uint32_t* sums)
{
    __shared__ uint32_t sh[kT / 64];
    const uint32_t gt = blockIdx.x * kT + threadIdx.x, b0 = gt * kBPT;
    uint32_t sum = 0, nib[kBPT];
#pragma unroll
    for (int j = 0; j < kBPT; j++) {
        uint32_t nb = 0;
        if (b0 + j < nblocks) nb = block_width(*(const v4*)(x + 8 * (uint64_t)(b0 + j)), zig);
        sum += nb;
        nib[j] = nb - (nb == 16u);
    }
#pragma unroll
    for (int k = 0; k < kBPT / 2; k++)
        if (gt * (kBPT / 2) + k < hdr_bytes) hdr[gt * (kBPT / 2) + k] = (uint8_t)(nib[2 * k] | (nib[2 * k + 1] << 4));
    const uint32_t total = workgroup_sum(sum, sh);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ void __launch_bounds__(kT) unpack_tile_kernel(const uint8_t* hdr, uint32_t nblocks, uint32_t* sums)
{
    __shared__ uint32_t sh[kT / 64];
    const uint32_t b0 = (blockIdx.x * kT + threadIdx.x) * kBPT;
    uint32_t sum = 0;
#pragma unroll
    for (int j = 0; j < kBPT; j++)
        if (b0 + j < nblocks) { const uint32_t nb = (hdr[(b0 + j) / 2] >> (4 * ((b0 + j) & 1u))) & 15u; sum += nb + (nb == 15u); }   // 15 means 16 (online.cpp:560)
    const uint32_t total = workgroup_sum(sum, sh);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// one workgroup: offs[i] = sum of sums[0 .. i-1], offs[ntiles] = everything.  The sums pass through LDS in slabs (read coalesced; a
// thread's consecutive ones are then LDS reads -- as dependent global loads this kernel was 20 us of a 100 us call).
// (Round 5: 1 024 threads x 8 sums instead of 256 x 32 -- the kernel is one workgroup's serial phases, 15.4 us of a 102 us unpack.)
constexpr int kScanT = 1024, kSumSlab = 8;
__global__ void __launch_bounds__(kScanT) tile_offsets_kernel(const uint32_t* sums, uint32_t ntiles, uint64_t* offs)
{
    __shared__ uint32_t slab[kScanT * kSumSlab];
    __shared__ uint32_t sh[kScanT / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint64_t carry = 0;
    for (uint32_t base = 0; base < ntiles; base += kScanT * kSumSlab) {
        for (int k = 0; k < kSumSlab; k++) {
            const uint32_t i = base + k * kScanT + threadIdx.x;
            slab[k * kScanT + threadIdx.x] = i < ntiles ? sums[i] : 0u;
        }
        __syncthreads();
        uint32_t mine = 0;
        for (int k = 0; k < kSumSlab; k++) mine += slab[threadIdx.x * kSumSlab + k];      // (a slab's payload is < 2^32: 8 192 tiles of <= 16 KB)
        uint32_t inc = mine;                                                              // inclusive scan over the wave, then over the 16 waves
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)inc, d); if (lane >= d) inc += u; }
        if (lane == 63) sh[w] = inc;
        __syncthreads();
        uint32_t wbase = 0, slab_total = 0;
#pragma unroll
        for (int k = 0; k < kScanT / 64; k++) { wbase += k < w ? sh[k] : 0u; slab_total += sh[k]; }
        uint32_t run = wbase + inc - mine;
        for (int k = 0; k < kSumSlab; k++) {
            const uint32_t v = slab[threadIdx.x * kSumSlab + k];
            slab[threadIdx.x * kSumSlab + k] = run;
            run += v;
        }
        __syncthreads();
        for (int k = 0; k < kSumSlab; k++) {
            const uint32_t i = base + k * kScanT + threadIdx.x;
            if (i < ntiles) offs[i] = carry + slab[k * kScanT + threadIdx.x];
        }
        carry += slab_total;
        __syncthreads();
    }
    if (threadIdx.x == 0) offs[ntiles] = carry;
}

constexpr uint32_t kPackImage = kTile * 16 + 32;                  // a tile's payload (<= 16 bytes a block) + its misalignment + slack

// The tile's payload is assembled in LDS (fields OR-ed into a zeroed image that starts on the 16-byte line the tile's first byte lies
// in) and leaves in 16-byte pieces; the first and last partial pieces go out byte by byte (the neighbours' bytes share those lines).
__global__ void __launch_bounds__(kT) pack_write_kernel(const uint16_t* x, uint32_t len, uint32_t nblocks, int zig, const uint64_t* tile_offs, uint32_t ntiles,
                                                        uint8_t* payload, uint32_t hdr_elems, int64_t* ret)
{
    __shared__ __attribute__((aligned(16))) uint32_t img[kPackImage / 4];
    __shared__ uint32_t sh[kT / 64];
    const uint32_t tile = blockIdx.x;
    const uint32_t t = threadIdx.x, b0 = (tile * kT + t) * kBPT;
    const uint64_t g0 = tile_offs[tile];
    const uint32_t tile_bytes = (uint32_t)(tile_offs[tile + 1] - g0);
    const uint32_t mis = (uint32_t)((uintptr_t)(payload + g0) & 15u);
    for (uint32_t u = t; u < (mis + tile_bytes + 31u) / 16u; u += kT) ((uint4*)img)[u] = make_uint4(0, 0, 0, 0);
    v4 q[kBPT];
    uint32_t nb[kBPT], sum = 0;
#pragma unroll
    for (int j = 0; j < kBPT; j++) {
        nb[j] = 0;
        q[j] = v4{0, 0, 0, 0};
        if (b0 + j < nblocks) {
            q[j] = *(const v4*)(x + 8 * (uint64_t)(b0 + j));
            nb[j] = block_width(q[j], zig);
            if (zig) { q[j].x = zz16x2(q[j].x); q[j].y = zz16x2(q[j].y); q[j].z = zz16x2(q[j].z); q[j].w = zz16x2(q[j].w); }
        }
        sum += nb[j];
    }
    uint32_t at = mis + workgroup_excl(sum, sh);                  // (its barriers also order the zeroing before the ORs)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)img;
#pragma unroll
    for (int j = 0; j < kBPT; j++) {
        const uint32_t n = nb[j];
        if (n) {
            // 8 fields of n bits = n bytes: the low four in one 64-bit word, the high four in another, joined 4 n bits up
            const uint64_t lo = (uint64_t)(q[j].x & 0xffffu) | ((uint64_t)(q[j].x >> 16) << n) | ((uint64_t)(q[j].y & 0xffffu) << (2 * n)) | ((uint64_t)(q[j].y >> 16) << (3 * n));
            const uint64_t hi = (uint64_t)(q[j].z & 0xffffu) | ((uint64_t)(q[j].z >> 16) << n) | ((uint64_t)(q[j].w & 0xffffu) << (2 * n)) | ((uint64_t)(q[j].w >> 16) << (3 * n));
            const uint32_t s = 4 * n;                              // 4 .. 64
            const uint64_t w01 = s < 64 ? lo | (hi << s) : lo;
            const uint64_t w23 = s < 64 ? hi >> (64 - s) : hi;
            const uint32_t a[6] = {0u, (uint32_t)w01, (uint32_t)(w01 >> 32), (uint32_t)w23, (uint32_t)(w23 >> 32), 0u};
            const uint32_t k = at & 3u;                            // byte shift inside the first dword
#pragma unroll
            for (int i = 0; i < 5; i++) {
                const uint32_t d = k ? __builtin_amdgcn_alignbyte(a[i + 1], a[i], 4u - k) : a[i + 1];
                if (4u * i < n + k && d)
                    __hip_atomic_fetch_or((__attribute__((address_space(3))) uint32_t*)(uintptr_t)(lds0 + (at & ~3u) + 4u * i), d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        at += n;
    }
    __syncthreads();
    // image byte i is payload byte (g0 - mis) + i
    uint8_t* const gbase = payload + g0 - mis;
    const uint32_t end = mis + tile_bytes;
    const uint8_t* const ib = (const uint8_t*)img;
    for (uint32_t u = t * 16u; u < end; u += kT * 16u) {
        if (u >= mis && u + 16u <= end) *(uint4*)(gbase + u) = *(const uint4*)(ib + u);
        else for (uint32_t k = u < mis ? mis : u; k < u + 16u && k < end; k++) gbase[k] = ib[k];
    }
    if (tile == ntiles - 1 && t == 0) {                           // the last tile knows where the payload of the full blocks ends
        const uint64_t pos = g0 + tile_bytes;
        uint8_t* o = payload + pos;
        const uint32_t tail = len - 8 * nblocks;
        for (uint32_t k = 0; k < tail; k++) st16(o + 2 * k, x[8 * (uint64_t)nblocks + k]);   // raw, at any byte alignment (:476-478)
        const uint64_t endb = pos + 2 * (uint64_t)tail;
        if (endb & 1) o[2 * tail] = 0;                           // the container is counted in elements: a defined pad byte
        if (ret) *ret = 2 + (int64_t)hdr_elems + (int64_t)((endb + 1) / 2);
    }
}

// the tile's payload -> LDS in 16-byte pieces (from the line its first byte lies in), then every thread takes its blocks from there
__global__ void __launch_bounds__(kT) unpack_read_kernel(const uint8_t* hdr, const uint8_t* payload, uint32_t len, uint32_t nblocks, int zig,
                                                         const uint64_t* tile_offs, uint32_t ntiles, uint16_t* out, int64_t* ret)
{
    __shared__ __attribute__((aligned(16))) uint32_t img[kPackImage / 4 + 4];
    __shared__ uint32_t sh[kT / 64];
    const uint32_t tile = blockIdx.x;
    const uint32_t t = threadIdx.x, b0 = (tile * kT + t) * kBPT;
    const uint64_t g0 = tile_offs[tile];
    const uint32_t tile_bytes = (uint32_t)(tile_offs[tile + 1] - g0);
    const uint32_t mis = (uint32_t)((uintptr_t)(payload + g0) & 15u);
    const uint8_t* const gbase = payload + g0 - mis;
    // (the last piece may reach past the payload by < 16 bytes: inside its own 16-byte line)
    for (uint32_t u = t * 16u; u < mis + tile_bytes; u += kT * 16u) *(uint4*)((uint8_t*)img + u) = *(const uint4*)(gbase + u);
    uint32_t nb[kBPT], sum = 0;
#pragma unroll
    for (int j = 0; j < kBPT; j++) {
        nb[j] = 0;
        if (b0 + j < nblocks) { const uint32_t f = (hdr[(b0 + j) / 2] >> (4 * ((b0 + j) & 1u))) & 15u; nb[j] = f + (f == 15u); }   // 15 means 16 (online.cpp:560)
        sum += nb[j];
    }
    const uint32_t excl = workgroup_excl(sum, sh);                // (its barriers also order the image's writes before the reads)
    uint32_t at = mis + excl;
#pragma unroll
    for (int j = 0; j < kBPT; j++) {
        const uint32_t n = nb[j];
        if (b0 + j < nblocks) {
            const uint32_t* const p = img + (at >> 2);
            const uint32_t k = at & 3u;
            const uint32_t a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3], a4 = p[4];
            const uint32_t w0 = __builtin_amdgcn_alignbyte(a1, a0, k), w1 = __builtin_amdgcn_alignbyte(a2, a1, k);
            const uint32_t w2 = __builtin_amdgcn_alignbyte(a3, a2, k), w3 = __builtin_amdgcn_alignbyte(a4, a3, k);
            const uint64_t w01 = ((uint64_t)w1 << 32) | w0, w23 = ((uint64_t)w3 << 32) | w2;
            const uint32_t s = 4 * n;                              // 0 .. 64
            const uint64_t lo = w01;
            const uint64_t hi = s == 0 ? 0ull : s < 64 ? (w01 >> s) | (w23 << (64 - s)) : w23;
            const uint32_t mask = n >= 16 ? 0xffffu : (1u << n) - 1u;
            uint32_t v[8];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                v[i] = (uint32_t)(lo >> (i * n)) & mask;
                v[4 + i] = (uint32_t)(hi >> (i * n)) & mask;
            }
            v4 o;
            o.x = v[0] | (v[1] << 16); o.y = v[2] | (v[3] << 16); o.z = v[4] | (v[5] << 16); o.w = v[6] | (v[7] << 16);
            if (zig) { o.x = unzz16x2(o.x); o.y = unzz16x2(o.y); o.z = unzz16x2(o.z); o.w = unzz16x2(o.w); }
            *(v4*)(out + 8 * (uint64_t)(b0 + j)) = o;
        }
        at += n;
    }
    if (tile == ntiles - 1 && t == 0) {
        const uint8_t* in = payload + g0 + tile_bytes;
        for (uint32_t k = 8 * nblocks; k < len; k++) out[k] = (uint16_t)ld16(in + 2 * (uint64_t)(k - 8 * nblocks));
        if (ret) *ret = checked_len(hdr, len);
    }
}

__global__ void header_kernel(uint8_t* dest, uint32_t len, uint8_t* zero_from, uint32_t zero_bytes, int64_t* ret, int64_t ret_value, int set_ret)
{
    if (threadIdx.x == 0) {
        dest[0] = (uint8_t)len; dest[1] = (uint8_t)(len >> 8); dest[2] = (uint8_t)(len >> 16); dest[3] = (uint8_t)(len >> 24);
        if (set_ret && ret) *ret = ret_value;
    }
    for (uint32_t k = threadIdx.x; k < zero_bytes; k += blockDim.x) zero_from[k] = 0;
}

__global__ void check_len_kernel(const uint8_t* src, uint32_t len, int64_t* ret)
{
    const uint32_t have = (uint32_t)src[0] | ((uint32_t)src[1] << 8) | ((uint32_t)src[2] << 16) | ((uint32_t)src[3] << 24);
    if (have != len && ret) *ret = SPRINTZ_E_CORRUPT;
}

int fail(int code, const char* what) { return sprintz::set_error(code, what); }
unsigned grid_for(uint64_t items) { return (unsigned)((items + kT - 1) / kT); }
size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }
unsigned tiles_for(uint64_t blocks) { return (unsigned)((blocks + kTile - 1) / kTile); }

// the one-pass dynamic-delta decoder: one workgroup a CU; SPRINTZ_MI355X_ONLINE_CHAIN: 0 = the three-launch form always, n > 0 = one pass from n
// tiles (of 8 192 blocks) on -- default 128; 1 lets the tests drive it with short streams
int dc_resident_wgs()
{
    static const int n = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        return cus * DC_WGS_PER_CU;
    }();
    return n;
}
uint32_t dc_min_tiles()
{
    const char* e = getenv("SPRINTZ_MI355X_ONLINE_CHAIN");
    if (!e || !e[0]) return 128u;            // (16 MB of samples: 25.1 against 26.3 us there, 17 - 19 against 15 below, 94 against 114 at 128 MB -- tools/chain_sizes.py)
    const long v = strtol(e, nullptr, 10);
    return v <= 0 ? 0xffffffffu : (uint32_t)v;
}

uint32_t choice_bytes_of(uint32_t len) { return (((len + 7) / 8) + 7) / 8; }                 // online.cpp:253-258
uint32_t hdr_bytes_of(uint32_t len) { return (((len + 7) / 8) * 4 + 7) / 8; }                // online.cpp:355-359

}  // namespace

namespace {
constexpr size_t kOnlinePinMax = 4u << 20;     // larger transfers go straight from/to the caller's memory

// `in_bytes` of `in` -> device, run(d_in, d_out, d_ret, d_tmp, stream), then ret elements of 2 bytes -> `out`
template <typename F>
int64_t online_host_call(const void* in, size_t in_bytes, size_t out_cap_bytes, size_t tmpb, void* out, F run)
{
    const size_t o_out = up256(in_bytes + 64), o_tmp = o_out + up256(out_cap_bytes + 64), o_ret = o_tmp + up256(tmpb + 64);
    const bool pin_in = in_bytes <= kOnlinePinMax, pin_out = out_cap_bytes <= kOnlinePinMax;
    HostScratch sc;
    int rc = host_scratch(o_ret + 16, std::max<size_t>(16, std::max(pin_in ? in_bytes : 0, pin_out ? out_cap_bytes : 0)), &sc);
    if (rc) return rc;
    const void* h_in = in;
    if (pin_in) { memcpy(sc.pin, in, in_bytes); h_in = sc.pin; }
    if (hipMemcpyAsync(sc.dev, h_in, in_bytes, hipMemcpyHostToDevice, sc.stream) != hipSuccess) return fail(SPRINTZ_E_HIP, "online: H2D copy");
    int64_t* d_ret = (int64_t*)(sc.dev + o_ret);
    rc = run(sc.dev, sc.dev + o_out, d_ret, sc.dev + o_tmp, sc.stream);
    if (rc) { (void)hipStreamSynchronize(sc.stream); return rc; }
    int64_t ret = SPRINTZ_E_HIP;
    // sc.pin is free again once the H2D copy has run; the 8 bytes of ret come back first, then exactly ret elements
    if (hipMemcpyAsync(sc.pin, d_ret, 8, hipMemcpyDeviceToHost, sc.stream) != hipSuccess || hipStreamSynchronize(sc.stream) != hipSuccess)
        return fail(SPRINTZ_E_HIP, "online: device call failed");
    memcpy(&ret, sc.pin, 8);
    if (ret > 0) {
        const size_t nb = (size_t)ret * 2;
        if (nb > out_cap_bytes) return fail(SPRINTZ_E_HIP, "online: the device reported more than the bound");
        void* h_out = pin_out ? (void*)sc.pin : out;
        if (hipMemcpyAsync(h_out, sc.dev + o_out, nb, hipMemcpyDeviceToHost, sc.stream) != hipSuccess || hipStreamSynchronize(sc.stream) != hipSuccess)
            return fail(SPRINTZ_E_HIP, "online: D2H copy");
        if (pin_out) memcpy(out, sc.pin, nb);
    }
    return ret;
}
}  // namespace

extern "C" {

size_t sprintz_mi355x_online_bound(int kind, uint32_t len)
{
    (void)kind;
    return (size_t)4 + (size_t)len * 2 + ((size_t)len + 7) / 8 + 32;
}

size_t sprintz_mi355x_online_tmp_bytes(int kind, uint32_t len)
{
    const uint64_t nblocks = len / 8 + 1;
    const uint64_t ntiles = (nblocks + 4 + kTile - 1) / kTile + 1;
    if (kind <= SPRINTZ_ONLINE_DYNDELTA_ALT) return (size_t)(ntiles * 8 + 256);               // one composed map a tile
    if (kind == SPRINTZ_ONLINE_ZIGZAG) return 16;
    return (size_t)(up256(ntiles * 4) + (ntiles + 1) * 8 + 256);                               // a payload size and an offset a tile
}

int sprintz_mi355x_online_pack_device(int kind, const uint16_t* d_src, uint32_t len, void* d_dest, int64_t* d_ret, void* d_tmp, void* hip_stream)
{
    if (kind < SPRINTZ_ONLINE_DYNDELTA || kind > SPRINTZ_ONLINE_PACK_ZIGZAG) return fail(SPRINTZ_E_INVALID, "online: kind must be 0..4");
    if (!d_dest || (len && !d_src) || !d_tmp || ((uintptr_t)d_dest & 15) || ((uintptr_t)d_src & 15)) return fail(SPRINTZ_E_INVALID, "online: null or misaligned (16-byte) device pointer");
    if (!have_device()) return fail(SPRINTZ_E_NO_DEVICE, "online: no usable HIP device (there is no CPU fallback)");
    hipStream_t st = (hipStream_t)hip_stream;
    uint8_t* dest = (uint8_t*)d_dest;
    uint8_t* body = dest + 4;
    if (kind == SPRINTZ_ONLINE_ZIGZAG) {
        hipLaunchKernelGGL(header_kernel, dim3(1), dim3(64), 0, st, dest, len, dest, 0u, d_ret, (int64_t)2 + len, 1);
        if (len) hipLaunchKernelGGL(zigzag_kernel, dim3(grid_for(((uint64_t)len + 7) / 8)), dim3(kT), 0, st, (const uint8_t*)d_src, body, len, 0, (int64_t*)nullptr);
    } else if (kind <= SPRINTZ_ONLINE_DYNDELTA_ALT) {
        const uint32_t cb = choice_bytes_of(len), celems = (cb + 1) / 2;
        uint8_t* choices = body + 2 * (size_t)len;
        // header, the choice bytes zeroed (bits of blocks that do not exist, the pad byte), and the return value of the short inputs
        // header; the return value of the short inputs.  The choice bytes (bits of blocks that do not exist and the pad byte included) are
        // written by the encode launch itself, whose grid reaches 64 blocks = 8 bytes past the last block (round 4 zeroed them here, a
        // megabyte for 64 Mi samples, with ONE workgroup: a third of the call)
        hipLaunchKernelGGL(header_kernel, dim3(1), dim3(256), 0, st, dest, len, choices, len >= 2 ? 0u : celems * 2, d_ret, (int64_t)2 + len + celems, 1);
        if (len == 1) (void)hipMemcpyAsync(body, d_src, 2, hipMemcpyDeviceToDevice, st);
        if (len >= 2) {
            const uint32_t nblocks = (len - 1) / 8;
            hipLaunchKernelGGL(dyndelta_encode_kernel, dim3(grid_for((uint64_t)nblocks + 64)), dim3(kT), 0, st, d_src, len, body, choices, celems * 2,
                               kind == SPRINTZ_ONLINE_DYNDELTA_ALT ? 1 : 0, d_ret);
        }
    } else {
        const int zig = kind == SPRINTZ_ONLINE_PACK_ZIGZAG;
        const uint32_t nblocks = len / 8, hb = hdr_bytes_of(len), helems = (hb + 1) / 2;
        uint8_t* hdr = body;
        uint8_t* payload = body + 2 * (size_t)helems;
        const uint32_t ntiles = tiles_for((uint64_t)nblocks + 4);          // (+ 4: the header's pad nibbles are zeroed by the threads behind the last block)
        uint32_t* sums = (uint32_t*)d_tmp;
        uint64_t* toffs = (uint64_t*)((uint8_t*)d_tmp + up256((size_t)ntiles * 4));
        hipLaunchKernelGGL(header_kernel, dim3(1), dim3(256), 0, st, dest, len, hdr, nblocks ? 0u : helems * 2, d_ret, (int64_t)2 + helems + len, nblocks ? 0 : 1);
        if (nblocks) {
            hipLaunchKernelGGL(pack_tile_kernel, dim3(ntiles), dim3(kT), 0, st, d_src, nblocks, zig, hdr, helems * 2, sums);
            hipLaunchKernelGGL(tile_offsets_kernel, dim3(1), dim3(kScanT), 0, st, (const uint32_t*)sums, ntiles, toffs);
            hipLaunchKernelGGL(pack_write_kernel, dim3(ntiles), dim3(kT), 0, st, d_src, len, nblocks, zig, (const uint64_t*)toffs, ntiles, payload, helems, d_ret);
        } else if (len) {                                           // fewer than 8 values: all of them raw
            (void)hipMemcpyAsync(payload, d_src, (size_t)len * 2, hipMemcpyDeviceToDevice, st);
        }
    }
    return hipGetLastError() == hipSuccess ? 0 : fail(SPRINTZ_E_HIP, "online: pack launch");
}

int sprintz_mi355x_online_unpack_device(int kind, const void* d_src, uint32_t len, uint16_t* d_dest, int64_t* d_ret, void* d_tmp, void* hip_stream)
{
    if (kind < SPRINTZ_ONLINE_DYNDELTA || kind > SPRINTZ_ONLINE_PACK_ZIGZAG) return fail(SPRINTZ_E_INVALID, "online: kind must be 0..4");
    if (!d_src || (len && !d_dest) || !d_tmp || ((uintptr_t)d_src & 15) || ((uintptr_t)d_dest & 15)) return fail(SPRINTZ_E_INVALID, "online: null or misaligned (16-byte) device pointer");
    if (!have_device()) return fail(SPRINTZ_E_NO_DEVICE, "online: no usable HIP device (there is no CPU fallback)");
    hipStream_t st = (hipStream_t)hip_stream;
    const uint8_t* src = (const uint8_t*)d_src;
    const uint8_t* body = src + 4;
    if (len == 0) {
        if (d_ret) (void)hipMemsetAsync(d_ret, 0, 8, st);
        hipLaunchKernelGGL(check_len_kernel, dim3(1), dim3(1), 0, st, src, len, d_ret);
        return 0;
    }
    if (kind == SPRINTZ_ONLINE_ZIGZAG) {
        hipLaunchKernelGGL(zigzag_kernel, dim3(grid_for(((uint64_t)len + 7) / 8)), dim3(kT), 0, st, body, (uint8_t*)d_dest, len, 1, d_ret);
    } else if (kind <= SPRINTZ_ONLINE_DYNDELTA_ALT) {
        const uint32_t nblocks = (len - 1) / 8, ntiles = (uint32_t)(((uint64_t)(nblocks ? nblocks : 1) + kDdTile - 1) / kDdTile);
        const uint8_t* choices = body + 2 * (size_t)len;
        uint64_t* tiles = (uint64_t*)d_tmp;
        // long streams: one pass (dyndelta_chain_kernel); its words fit the scratch of the three-launch form (a word per 8 192 blocks + the ticket)
        const uint32_t ctiles = (uint32_t)(((uint64_t)nblocks + kDcTileBlocks - 1) / kDcTileBlocks);
        if (ctiles >= dc_min_tiles()) {
            if (hipMemsetAsync(d_tmp, 0, 256 + (size_t)ctiles * 8, st) != hipSuccess) return fail(SPRINTZ_E_HIP, "online: hipMemsetAsync of the tiles' words");
            const uint32_t wgs = ctiles < (uint32_t)dc_resident_wgs() ? ctiles : (uint32_t)dc_resident_wgs();
            hipLaunchKernelGGL(dyndelta_chain_kernel, dim3(wgs), dim3(kDcT), 0, st, body, choices, len, nblocks, ctiles, (uint32_t*)d_tmp,
                               (uint64_t*)((uint8_t*)d_tmp + 256), d_dest, d_ret);
            return hipGetLastError() == hipSuccess ? 0 : fail(SPRINTZ_E_HIP, "online: unpack launch");
        }
        hipLaunchKernelGGL(dyndelta_tile_kernel, dim3(ntiles), dim3(kT), 0, st, body, choices, nblocks, tiles);
        hipLaunchKernelGGL(dyndelta_tilescan_kernel, dim3(1), dim3(kDdScanT), 0, st, tiles, ntiles);
        hipLaunchKernelGGL(dyndelta_decode_kernel, dim3(ntiles), dim3(kT), 0, st, body, choices, len, nblocks, (const uint64_t*)tiles, d_dest, d_ret);
    } else {
        const int zig = kind == SPRINTZ_ONLINE_PACK_ZIGZAG;
        const uint32_t nblocks = len / 8, hb = hdr_bytes_of(len), helems = (hb + 1) / 2;
        const uint8_t* payload = body + 2 * (size_t)helems;
        const uint32_t ntiles = tiles_for(nblocks ? nblocks : 1);
        uint32_t* sums = (uint32_t*)d_tmp;
        uint64_t* toffs = (uint64_t*)((uint8_t*)d_tmp + up256((size_t)ntiles * 4));
        hipLaunchKernelGGL(unpack_tile_kernel, dim3(ntiles), dim3(kT), 0, st, body, nblocks, sums);
        hipLaunchKernelGGL(tile_offsets_kernel, dim3(1), dim3(kScanT), 0, st, (const uint32_t*)sums, ntiles, toffs);
        hipLaunchKernelGGL(unpack_read_kernel, dim3(ntiles), dim3(kT), 0, st, body, payload, len, nblocks, zig, (const uint64_t*)toffs, ntiles, d_dest, d_ret);
    }
    // (the header must agree with the caller: checked_len(), by the thread of the last launch that writes the return value)
    return hipGetLastError() == hipSuccess ? 0 : fail(SPRINTZ_E_HIP, "online: unpack launch");
}

// ---- single-call forms over host buffers (the reference's signatures are in include/sprintz_dropin.hpp)
// Like api.hip's single calls: the thread's pooled scratch (one device buffer [source | destination | tmp | ret], one
// pinned staging buffer, one private non-blocking stream) -- no hipMalloc / hipFree per call, no device-wide
// synchronisation that would stall other threads' streams.

int64_t sprintz_mi355x_online_pack(int kind, const uint16_t* src, uint32_t len, int16_t* dest)
{
    if (kind < SPRINTZ_ONLINE_DYNDELTA || kind > SPRINTZ_ONLINE_PACK_ZIGZAG) return fail(SPRINTZ_E_INVALID, "online: kind must be 0..4");
    if (!dest || (len && !src)) return fail(SPRINTZ_E_INVALID, "online: null pointer");
    if (len > (1u << 30)) return fail(SPRINTZ_E_UNSUPPORTED, "online: single call limited to 2^30 elements");
    if (!have_device()) return fail(SPRINTZ_E_NO_DEVICE, "online: no usable HIP device (there is no CPU fallback)");
    const size_t bound = sprintz_mi355x_online_bound(kind, len), tmpb = sprintz_mi355x_online_tmp_bytes(kind, len);
    return online_host_call(src, (size_t)len * 2, bound, tmpb, dest, [&](uint8_t* d_in, uint8_t* d_out, int64_t* d_ret, uint8_t* d_tmp, hipStream_t st) {
        return sprintz_mi355x_online_pack_device(kind, (const uint16_t*)d_in, len, d_out, d_ret, d_tmp, st);
    });
}

int64_t sprintz_mi355x_online_unpack(int kind, const int16_t* src, uint16_t* dest)
{
    if (kind < SPRINTZ_ONLINE_DYNDELTA || kind > SPRINTZ_ONLINE_PACK_ZIGZAG) return fail(SPRINTZ_E_INVALID, "online: kind must be 0..4");
    if (!src || !dest) return fail(SPRINTZ_E_INVALID, "online: null pointer");
    uint32_t len;
    memcpy(&len, src, 4);                                          // read_metadata_simple1d (format.h:95-99)
    if (len > (1u << 30)) return fail(SPRINTZ_E_UNSUPPORTED, "online: single call limited to 2^30 elements (damaged header?)");
    if (len == 0) return 0;
    if (!have_device()) return fail(SPRINTZ_E_NO_DEVICE, "online: no usable HIP device (there is no CPU fallback)");
    const size_t tmpb = sprintz_mi355x_online_tmp_bytes(kind, len);
    // the caller's buffer holds the container and not necessarily a byte more: a dynamic-delta / zigzag container's size follows
    // from len; a sprintzpack container's from its header nibbles (host framing walk, no sample touched)
    size_t have = 4 + (size_t)len * 2;
    if (kind <= SPRINTZ_ONLINE_DYNDELTA_ALT) have += (size_t)((choice_bytes_of(len) + 1) / 2) * 2;
    if (kind >= SPRINTZ_ONLINE_PACK) {
        const uint8_t* h = (const uint8_t*)src + 4;
        const uint32_t nblocks = len / 8, helems = (hdr_bytes_of(len) + 1) / 2;
        size_t pay = 0;
        for (uint32_t b = 0; b < nblocks; b++) { uint32_t nb = (h[b / 2] >> (4 * (b & 1u))) & 15u; pay += nb + (nb == 15u); }
        have = 4 + (size_t)helems * 2 + pay + (size_t)(len - 8 * nblocks) * 2;
    }
    return online_host_call(src, have, (size_t)len * 2, tmpb, dest, [&](uint8_t* d_in, uint8_t* d_out, int64_t* d_ret, uint8_t* d_tmp, hipStream_t st) {
        return sprintz_mi355x_online_unpack_device(kind, d_in, len, (uint16_t*)d_out, d_ret, d_tmp, st);
    });
}

}  // extern "C"

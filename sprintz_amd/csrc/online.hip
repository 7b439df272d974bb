// online.hip -- the reference's 2020 "online" coders for 1-D uint16 streams (cpp/Compress/online.hpp:395-445,
// online.cpp) on gfx950:
//   dynamic_delta_pack_u16 / _altloss / dynamic_delta_unpack_u16     online.cpp:48-311
//   zigzag_pack_u16 / zigzag_unpack_u16                               online.cpp:314-351
//   sprintzpack_pack_u16 / _zigzag / sprintzpack_unpack_u16 / _zigzag online.cpp:355-703
// Container formats and every quirk are restated, with citations, in oracle/online_oracle.c (pinned against the
// compiled reference).  One call codes ONE stream of any length; the reference walks it serially, here:
//   * encoders are block-parallel: a block's predictor choice / bit width is a pure function of the input (both of the
//     reference's predictors are trained on the true values, so neither depends on earlier choices);
//   * sprintzpack's byte offsets are an exclusive scan of the per-block widths (launch_size_scan, api.hip);
//   * the dynamic-delta DECODER is a scan too: a block maps the running state (x, d) = (last value, last difference)
//     affinely -- delta: (x + A, e7); double delta: (x + 8 d + C, d + A) with A = sum e, C = sum (8 - i) e_i -- and such
//     maps compose associatively as (m, a, tx, td): x' = x + m d + tx, d' = a d + td, everything modulo 2^16.
//     Three launches: tile summaries (256 blocks = 2048 samples per workgroup), one workgroup scanning the tiles,
//     and the decode proper with every thread's incoming state.
#include "../../include/sprintz_mi355x.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>

#include "launch.h"

using namespace sprintz;

namespace {

constexpr int kT = 256;                                   // threads per workgroup = blocks of 8 samples per tile

__device__ __forceinline__ uint32_t zz16(uint32_t x)      // zigzag_encode_16b of the int16 in the low half (bitpack.h:312)
{
    const int v = (int)(int16_t)x;
    return ((uint32_t)(v << 1) ^ (uint32_t)(v >> 15)) & 0xffffu;
}
__device__ __forceinline__ uint32_t unzz16(uint32_t z) { return ((z >> 1) ^ (0u - (z & 1u))) & 0xffffu; }   // bitpack.h:315

typedef uint16_t __attribute__((aligned(1), may_alias)) u16_a1;
typedef uint32_t __attribute__((aligned(1), may_alias)) u32_a1;

__device__ __forceinline__ uint32_t ld16(const uint8_t* p) { return *(const u16_a1*)p; }
__device__ __forceinline__ void st16(uint8_t* p, uint32_t v) { *(u16_a1*)p = (uint16_t)v; }

// ---------------------------------------------------------------- zigzag
__global__ void __launch_bounds__(kT) zigzag_kernel(const uint8_t* src, uint8_t* dst, uint32_t len, int decode, int64_t* ret)
{
    // src/dst are byte pointers to the first SAMPLE on both sides (2-byte aligned); 8 samples a thread
    const uint64_t i0 = ((uint64_t)blockIdx.x * kT + threadIdx.x) * 8;
    for (int i = 0; i < 8; i++) {
        const uint64_t i1 = i0 + i;
        if (i1 < len) st16(dst + 2 * i1, decode ? unzz16(ld16(src + 2 * i1)) : zz16(ld16(src + 2 * i1)));
    }
    if (i0 == 0 && ret) *ret = decode ? (int64_t)len : 2 + (int64_t)len;
}

// ---------------------------------------------------------------- dynamic delta, encoder
// thread = block b of 8 samples (elements 1 + 8 b ... 8 + 8 b); x: the samples; out: the container's sample area
__global__ void __launch_bounds__(kT) dyndelta_encode_kernel(const uint16_t* x, uint32_t len, uint8_t* out, uint8_t* choices,
                                                             uint32_t choice_bytes, int alt, int64_t* ret)
{
    const uint32_t n = len - 1, nblocks = n / 8;           // len >= 2 here
    const uint32_t b = blockIdx.x * kT + threadIdx.x;
    int choice = 0;
    if (b < nblocks) {
        const uint32_t at0 = 1 + 8 * b;
        uint32_t p1 = x[at0 - 1], p2 = at0 >= 2 ? x[at0 - 2] : x[0];
        uint32_t z0[8], z1[8], m0 = 0, m1 = 0, s0 = 0, s1 = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t v = x[at0 + i];
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
#pragma unroll
        for (int i = 0; i < 8; i++) st16(out + 2 * (uint64_t)(at0 + i), choice ? z1[i] : z0[i]);
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

// block b's map from its 8 zigzagged errors; e[] receives the errors
__device__ __forceinline__ Aff block_map(const uint8_t* in, const uint8_t* choices, uint32_t b, uint32_t (&e)[8], int& choice)
{
    choice = (choices[b / 8] >> (b % 8)) & 1;
    uint32_t A = 0, C = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        e[i] = unzz16(ld16(in + 2 * (uint64_t)(1 + 8 * b + i)));
        A += e[i];
        C += (uint32_t)(8 - i) * e[i];
    }
    return choice ? Aff{8u, 1u, C & 0xffffu, A & 0xffffu} : Aff{0u, 0u, A & 0xffffu, e[7]};
}

// inclusive scan of one map per thread over the workgroup (composition is not commutative: left operand = earlier)
__device__ Aff workgroup_scan(Aff mine, Aff& total, uint64_t* sh)
{
    const int t = threadIdx.x;
    sh[t] = pack_aff(mine);
    __syncthreads();
    for (int off = 1; off < kT; off <<= 1) {
        uint64_t prev = kIdentity;
        if (t >= off) prev = sh[t - off];
        __syncthreads();
        if (t >= off) sh[t] = pack_aff(compose(unpack_aff(prev), unpack_aff(sh[t])));
        __syncthreads();
    }
    total = unpack_aff(sh[kT - 1]);
    return unpack_aff(sh[t]);
}

__global__ void __launch_bounds__(kT) dyndelta_tile_kernel(const uint8_t* in, const uint8_t* choices, uint32_t nblocks, uint64_t* tiles)
{
    __shared__ uint64_t sh[kT];
    const uint32_t b = blockIdx.x * kT + threadIdx.x;
    Aff f = unpack_aff(kIdentity);
    if (b < nblocks) { uint32_t e[8]; int c; f = block_map(in, choices, b, e, c); }
    Aff total;
    (void)workgroup_scan(f, total, sh);
    if (threadIdx.x == 0) tiles[blockIdx.x] = pack_aff(total);
}

// one workgroup: tiles[i] <- composition of tiles[0 .. i-1] (exclusive)
__global__ void __launch_bounds__(kT) dyndelta_tilescan_kernel(uint64_t* tiles, uint32_t ntiles)
{
    __shared__ uint64_t sh[kT];
    const uint32_t per = (ntiles + kT - 1) / kT, lo = threadIdx.x * per, hi = lo + per < ntiles ? lo + per : ntiles;
    Aff mine = unpack_aff(kIdentity);
    for (uint32_t i = lo; i < hi; i++) mine = compose(mine, unpack_aff(tiles[i]));
    Aff total;
    const Aff incl = workgroup_scan(mine, total, sh);
    __syncthreads();
    // exclusive prefix of this thread's run = inclusive of the previous thread
    Aff run = threadIdx.x == 0 ? unpack_aff(kIdentity) : unpack_aff(sh[threadIdx.x - 1]);
    (void)incl;
    for (uint32_t i = lo; i < hi; i++) {
        const Aff t = unpack_aff(tiles[i]);
        tiles[i] = pack_aff(run);
        run = compose(run, t);
    }
}

__global__ void __launch_bounds__(kT) dyndelta_decode_kernel(const uint8_t* in, const uint8_t* choices, uint32_t len, uint32_t nblocks,
                                                             const uint64_t* tiles, uint16_t* out, int64_t* ret)
{
    __shared__ uint64_t sh[kT];
    const uint32_t b = blockIdx.x * kT + threadIdx.x;
    uint32_t e[8];
    int choice = 0;
    Aff f = unpack_aff(kIdentity);
    if (b < nblocks) f = block_map(in, choices, b, e, choice);
    Aff total;
    const Aff incl = workgroup_scan(f, total, sh);
    __syncthreads();
    // state before this thread's block: (x0, 0) through the tiles before, then the threads before
    const Aff before_tile = unpack_aff(tiles[blockIdx.x]);
    const Aff before = threadIdx.x == 0 ? before_tile : compose(before_tile, unpack_aff(sh[threadIdx.x - 1]));
    (void)incl;
    const uint32_t x0 = ld16(in);
    uint32_t x = (x0 + before.tx) & 0xffffu, d = before.td;       // d starts at 0 (online.hpp: _prev_diff = 0)
    if (b < nblocks) {
        uint32_t v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            d = choice ? (d + e[i]) & 0xffffu : e[i];
            x = (x + d) & 0xffffu;
            v[i] = x;
        }
#pragma unroll
        for (int i = 0; i < 8; i++) out[1 + 8 * (uint64_t)b + i] = (uint16_t)v[i];
    }
    if (b == (nblocks ? nblocks - 1 : 0)) {                       // the owner of the last block walks the < 8 trailing delta errors (:245-251)
        for (uint32_t at = 1 + 8 * nblocks; at < len; at++) {
            x = (x + ld16(in + 2 * (uint64_t)at)) & 0xffffu;
            out[at] = (uint16_t)x;
        }
    }
    if (b == 0) {
        out[0] = (uint16_t)x0;
        if (ret) *ret = len;
    }
}

// ---------------------------------------------------------------- sprintzpack
// thread = block of 8 values.  widths[b] = payload bytes of block b (= its bit width); header nibbles two blocks a byte.
__global__ void __launch_bounds__(kT) pack_width_kernel(const uint16_t* x, uint32_t nblocks, int zig, uint32_t* widths, uint8_t* hdr, uint32_t hdr_bytes)
{
    const uint32_t b = blockIdx.x * kT + threadIdx.x;
    uint32_t nb = 0;
    if (b < nblocks) {
        const uint4 q = *(const uint4*)(x + 8 * (uint64_t)b);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
        uint32_t all = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t lo = w[k] & 0xffffu, hi = w[k] >> 16;
            all |= zig ? (zz16(lo) | zz16(hi)) : (lo | hi);
        }
        nb = 32u - (uint32_t)__clz((int)all);
        nb = all == 0 ? 0u : nb;
        nb += nb == 15u;                                           // bitpack.h:286
        widths[b] = nb;
    }
    // nibble of block b: nb - (nb == 16); the odd lane hands its nibble to the even one (online.cpp:405-412)
    const uint32_t nib = nb - (nb == 16u);
    const uint32_t other = (uint32_t)__shfl_down((int)nib, 1);
    if ((b & 1u) == 0 && b / 2 < hdr_bytes) hdr[b / 2] = (uint8_t)(b < nblocks ? (nib | ((b + 1 < nblocks ? other : 0u) << 4)) : 0u);
}

__global__ void __launch_bounds__(kT) pack_write_kernel(const uint16_t* x, uint32_t len, uint32_t nblocks, int zig, const uint32_t* widths,
                                                        const uint64_t* offsets, uint8_t* payload, uint32_t hdr_elems, int64_t* ret)
{
    const uint32_t b = blockIdx.x * kT + threadIdx.x;
    if (b < nblocks) {
        const uint32_t nb = widths[b];
        uint8_t* o = payload + offsets[b];
        unsigned __int128 acc = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint32_t v = x[8 * (uint64_t)b + i];
            v = zig ? zz16(v) : v;
            acc |= (unsigned __int128)(v & ((1u << nb) - 1u)) << (i * nb);
        }
        for (uint32_t k = 0; k < nb; k++) o[k] = (uint8_t)(acc >> (8 * k));
    }
    if (b == 0) {
        const uint64_t pos = offsets[nblocks];                    // payload bytes of the full blocks
        uint8_t* o = payload + pos;
        const uint32_t tail = len - 8 * nblocks;
        for (uint32_t k = 0; k < tail; k++) st16(o + 2 * k, x[8 * (uint64_t)nblocks + k]);   // raw, at any byte alignment (:476-478)
        const uint64_t end = pos + 2 * (uint64_t)tail;
        if (end & 1) o[2 * tail] = 0;                            // the container is counted in elements: a defined pad byte
        if (ret) *ret = 2 + (int64_t)hdr_elems + (int64_t)((end + 1) / 2);
    }
}

__global__ void __launch_bounds__(kT) unpack_width_kernel(const uint8_t* hdr, uint32_t nblocks, uint32_t* widths)
{
    const uint32_t b = blockIdx.x * kT + threadIdx.x;
    if (b >= nblocks) return;
    uint32_t nb = (hdr[b / 2] >> (4 * (b & 1u))) & 15u;
    nb += nb == 15u;                                               // 15 means 16 (online.cpp:560)
    widths[b] = nb;
}

__global__ void __launch_bounds__(kT) unpack_read_kernel(const uint8_t* payload, uint32_t len, uint32_t nblocks, int zig, const uint32_t* widths,
                                                         const uint64_t* offsets, uint16_t* out, int64_t* ret)
{
    const uint32_t b = blockIdx.x * kT + threadIdx.x;
    if (b < nblocks) {
        const uint32_t nb = widths[b];
        const uint8_t* in = payload + offsets[b];
        unsigned __int128 acc = 0;
        for (uint32_t k = 0; k < nb; k++) acc |= (unsigned __int128)in[k] << (8 * k);
        uint32_t v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t z = (uint32_t)(acc >> (i * nb)) & ((1u << nb) - 1u);
            v[i] = zig ? unzz16(z) : z;
        }
        uint4 q;
        q.x = v[0] | (v[1] << 16); q.y = v[2] | (v[3] << 16); q.z = v[4] | (v[5] << 16); q.w = v[6] | (v[7] << 16);
        *(uint4*)(out + 8 * (uint64_t)b) = q;
    }
    if (b == 0) {
        const uint8_t* in = payload + offsets[nblocks];
        for (uint32_t at = 8 * nblocks; at < len; at++) out[at] = (uint16_t)ld16(in + 2 * (uint64_t)(at - 8 * nblocks));
        if (ret) *ret = len;
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

uint32_t choice_bytes_of(uint32_t len) { return (((len + 7) / 8) + 7) / 8; }                 // online.cpp:253-258
uint32_t hdr_bytes_of(uint32_t len) { return (((len + 7) / 8) * 4 + 7) / 8; }                // online.cpp:355-359

}  // namespace

namespace {
constexpr size_t kOnlinePinMax = 4u << 20;     // larger transfers go straight from/to the caller's memory
size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

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
    if (kind <= SPRINTZ_ONLINE_DYNDELTA_ALT) return (size_t)(((nblocks + kT - 1) / kT + 1) * 8 + 256);
    if (kind == SPRINTZ_ONLINE_ZIGZAG) return 16;
    return (size_t)(nblocks * 4 + 256 + (nblocks + 1) * 8 + 256 + sprintz_mi355x_compact_tmp_bytes(nblocks) + 256);
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
        hipLaunchKernelGGL(header_kernel, dim3(1), dim3(256), 0, st, dest, len, choices, celems * 2, d_ret, (int64_t)2 + len + celems, 1);
        if (len == 1) (void)hipMemcpyAsync(body, d_src, 2, hipMemcpyDeviceToDevice, st);
        if (len >= 2) {
            const uint32_t nblocks = (len - 1) / 8;
            hipLaunchKernelGGL(dyndelta_encode_kernel, dim3(grid_for(nblocks ? nblocks : 1)), dim3(kT), 0, st, d_src, len, body, choices, celems * 2,
                               kind == SPRINTZ_ONLINE_DYNDELTA_ALT ? 1 : 0, d_ret);
        }
    } else {
        const int zig = kind == SPRINTZ_ONLINE_PACK_ZIGZAG;
        const uint32_t nblocks = len / 8, hb = hdr_bytes_of(len), helems = (hb + 1) / 2;
        uint8_t* hdr = body;
        uint8_t* payload = body + 2 * (size_t)helems;
        uint32_t* widths = (uint32_t*)d_tmp;
        uint64_t* offsets = (uint64_t*)((uint8_t*)d_tmp + (((size_t)nblocks + 1) * 4 + 255) / 256 * 256);
        void* scan_tmp = (uint8_t*)offsets + (((size_t)nblocks + 2) * 8 + 255) / 256 * 256;
        hipLaunchKernelGGL(header_kernel, dim3(1), dim3(256), 0, st, dest, len, hdr, nblocks ? 0u : helems * 2, d_ret, (int64_t)2 + helems + len, nblocks ? 0 : 1);
        if (nblocks) {
            hipLaunchKernelGGL(pack_width_kernel, dim3(grid_for((uint64_t)nblocks + 4)), dim3(kT), 0, st, d_src, nblocks, zig, widths, hdr, helems * 2);
            if (launch_size_scan(widths, nblocks, 1, offsets, scan_tmp, st) != hipSuccess) return fail(SPRINTZ_E_HIP, "online: size scan launch");
            hipLaunchKernelGGL(pack_write_kernel, dim3(grid_for(nblocks)), dim3(kT), 0, st, d_src, len, nblocks, zig, widths, offsets, payload, helems, d_ret);
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
    if (d_ret) (void)hipMemsetAsync(d_ret, 0, 8, st);
    if (len == 0) { hipLaunchKernelGGL(check_len_kernel, dim3(1), dim3(1), 0, st, src, len, d_ret); return 0; }
    if (kind == SPRINTZ_ONLINE_ZIGZAG) {
        hipLaunchKernelGGL(zigzag_kernel, dim3(grid_for(((uint64_t)len + 7) / 8)), dim3(kT), 0, st, body, (uint8_t*)d_dest, len, 1, d_ret);
    } else if (kind <= SPRINTZ_ONLINE_DYNDELTA_ALT) {
        const uint32_t nblocks = (len - 1) / 8, ntiles = grid_for(nblocks ? nblocks : 1);
        const uint8_t* choices = body + 2 * (size_t)len;
        uint64_t* tiles = (uint64_t*)d_tmp;
        hipLaunchKernelGGL(dyndelta_tile_kernel, dim3(ntiles), dim3(kT), 0, st, body, choices, nblocks, tiles);
        hipLaunchKernelGGL(dyndelta_tilescan_kernel, dim3(1), dim3(kT), 0, st, tiles, ntiles);
        hipLaunchKernelGGL(dyndelta_decode_kernel, dim3(ntiles), dim3(kT), 0, st, body, choices, len, nblocks, (const uint64_t*)tiles, d_dest, d_ret);
    } else {
        const int zig = kind == SPRINTZ_ONLINE_PACK_ZIGZAG;
        const uint32_t nblocks = len / 8, hb = hdr_bytes_of(len), helems = (hb + 1) / 2;
        const uint8_t* payload = body + 2 * (size_t)helems;
        uint32_t* widths = (uint32_t*)d_tmp;
        uint64_t* offsets = (uint64_t*)((uint8_t*)d_tmp + (((size_t)nblocks + 1) * 4 + 255) / 256 * 256);
        void* scan_tmp = (uint8_t*)offsets + (((size_t)nblocks + 2) * 8 + 255) / 256 * 256;
        if (nblocks) {
            hipLaunchKernelGGL(unpack_width_kernel, dim3(grid_for(nblocks)), dim3(kT), 0, st, body, nblocks, widths);
            if (launch_size_scan(widths, nblocks, 1, offsets, scan_tmp, st) != hipSuccess) return fail(SPRINTZ_E_HIP, "online: size scan launch");
        } else {
            (void)hipMemsetAsync(offsets, 0, 8, st);
        }
        hipLaunchKernelGGL(unpack_read_kernel, dim3(grid_for(nblocks ? nblocks : 1)), dim3(kT), 0, st, payload, len, nblocks, zig, (const uint32_t*)widths,
                           (const uint64_t*)offsets, d_dest, d_ret);
    }
    hipLaunchKernelGGL(check_len_kernel, dim3(1), dim3(1), 0, st, src, len, d_ret);     // the header must agree with the caller
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

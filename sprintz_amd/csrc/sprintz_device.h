// sprintz_device.h -- device-side building blocks shared by the gfx950 kernels.
//
// Lane mapping used by every kernel in this library ("the generic mapping"):
// a chunk (independent compress() call) is owned by a GROUP of DP = 2^k
// adjacent lanes of one wavefront; lane `lane_d` of the group owns the CPL
// columns [lane_d*CPL, lane_d*CPL + CPL) that are < D.  64/DP chunks share a
// wavefront.  Everything that is sequential in the reference (stream position,
// RLE state, FIRE recurrence down a column) is lane-local or group-uniform;
// the only cross-lane traffic is the per-block nbits scan inside the group.
//
// Bit-exact semantics follow dblalock/sprintz cpp/Compress; citations inline.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sprintz {

template <int W> struct Elem;
template <> struct Elem<8> {
    using U = uint8_t;
    static constexpr int HB = 3;        // header field bits (sprintz_xff_rle.cpp:67)
    static constexpr uint32_t MASK = 0xffu;
};
template <> struct Elem<16> {
    using U = uint16_t;
    static constexpr int HB = 4;
    static constexpr uint32_t MASK = 0xffffu;
};

template <int W> __device__ __forceinline__ int sext(int x)
{
    if constexpr (W == 8) return (int)(int8_t)x;
    else return (int)(int16_t)x;
}

// zigzag at W bits (bitpack.h:302-303)
template <int W> __device__ __forceinline__ uint32_t zigzag(int e)
{
    return ((uint32_t)(e << 1) ^ (uint32_t)(e >> 31)) & Elem<W>::MASK;   // e is sign-extended
}
__device__ __forceinline__ int unzigzag(uint32_t z) { return (int)(z >> 1) ^ -(int)(z & 1u); }

// FIRE coefficient from the accumulator:
//   general layout  sprintz_xff_rle.cpp:217     int16((ctr >> (1+(W-4))) << (W-4))
//   low-dim layout  sprintz_xff_lowdim.cpp:170  ctr >> 1   (truncate_coeffs == false)
template <int W, bool LOWDIM> __device__ __forceinline__ int fire_coef(int ctr)
{
    if constexpr (LOWDIM) return ctr >> 1;
    else return (int)(int16_t)((uint32_t)(ctr >> (1 + (W - 4))) << (W - 4));
}

// The reference DECODER's coefficient while it replays a RUN of a 16-bit general-layout FIRE stream (sprintz_xff_rle.cpp:893-901):
// the 32-bit counters >> 13, read as 16-bit lanes -- an even column sees its low half, an odd column the HIGH half (the counters
// are not repositioned for the odd columns) -- shifted left by 4 instead of 12.  It does not invert the reference encoder; it is
// what a caller needs who must reproduce the reference decoder sample for sample (SPRINTZ_OPT_REF_DECODER_QUIRK).
__device__ __forceinline__ int fire_coef_ref_run16(int ctr, int column)
{
    const uint32_t k = (uint32_t)(ctr >> 13);
    const uint32_t lane = (column & 1) ? k >> 16 : k & 0xffffu;
    return (int)(int16_t)(uint16_t)(lane << 4);
}

// prediction = (prev_delta * coef) >> W truncated to W bits (sprintz_xff_rle.cpp:225).
// Both operands fit 24 bits except in the 16-bit low-dim codec, whose 32-bit
// coefficient needs the full (wrapping) multiply (sprintz_xff_lowdim.cpp:183).
template <int W, bool LOWDIM> __device__ __forceinline__ int fire_predict(int prev_delta, int coef)
{
    int prod;
    if constexpr (W == 16 && LOWDIM) prod = (int)((uint32_t)prev_delta * (uint32_t)coef);
    else prod = __mul24(prev_delta, coef);
    return sext<W>(prod >> W);
}

// FIRE counter wraps at int16 for 8-bit data, int32 for 16-bit (util.h:39-47)
template <int W> __device__ __forceinline__ int wrap_counter(int c)
{
    if constexpr (W == 8) return (int)(int16_t)c;
    else return c;
}

// icopysign(err, prev_delta) (util.h:63-68): sign(err) * prev_delta
__device__ __forceinline__ int sign_times(int err, int pd) { return err > 0 ? pd : (err < 0 ? -pd : 0); }

// nbits of a column from the OR of its zigzagged errors.
//   general: sprintz_xff_rle.cpp:259-265,284 with bitpack.h:72-93 (7->8 applied
//            to the high byte if it is non-zero, else to the low byte)
//   low-dim: sprintz_xff_lowdim.cpp:207-208 (only W-1 -> W)
template <int W, bool LOWDIM> __device__ __forceinline__ uint32_t nbits_of(uint32_t mask)
{
    uint32_t n = 32u - (uint32_t)__clz((int)mask);   // __clz(0) == 32
    if constexpr (LOWDIM) return n == (uint32_t)(W - 1) ? (uint32_t)W : n;
    else if constexpr (W == 8) return n == 7u ? 8u : n;
    else return (n == 7u || n == 15u) ? n + 1u : n;    // hi byte bitlen 7 <=> n == 15
}

// ---- unaligned, over-read-bounded fetches from a byte stream ---------------
// 32 bits starting at byte address p (any alignment): two aligned dwords + v_alignbit.
__device__ __forceinline__ uint32_t load_u32_any(const uint8_t* p)
{
    const uintptr_t a = (uintptr_t)p;
    const uint32_t* q = (const uint32_t*)(a & ~(uintptr_t)3);
    const uint32_t sh = ((uint32_t)a & 3u) << 3;
    return __builtin_amdgcn_alignbit(q[1], q[0], sh);
}
// n (<= 16) bits at bit offset `bitpos` from byte address base, LSB-first.
// Reads the aligned 8-byte window that contains them (<= 7 bytes of over-read).
__device__ __forceinline__ uint32_t fetch_bits(const uint8_t* base, uint32_t bitpos, uint32_t n)
{
    const uintptr_t a = (uintptr_t)(base + (bitpos >> 3));
    const uint32_t* q = (const uint32_t*)(a & ~(uintptr_t)3);
    const uint32_t sh = (((uint32_t)a & 3u) << 3) + (bitpos & 7u);      // <= 31
    const uint32_t w = __builtin_amdgcn_alignbit(q[1], q[0], sh);
    return __builtin_amdgcn_ubfe(w, 0, n);
}
__device__ __forceinline__ uint32_t load_u8(const uint8_t* p) { return *p; }

// ---- intra-group collectives (DP lanes, DP = 2^k, group aligned in the wave)
// exclusive prefix sum over the group; `total` = group sum
__device__ __forceinline__ uint32_t group_excl_scan(uint32_t v, int lane_d, int DP, uint32_t& total)
{
    uint32_t incl = v;
    for (int off = 1; off < DP; off <<= 1) {
        uint32_t t = __shfl_up(incl, off, DP);
        if (lane_d >= off) incl += t;
    }
    total = __shfl(incl, DP - 1, DP);
    return incl - v;
}
__device__ __forceinline__ uint32_t group_sum(uint32_t v, int DP)
{
    for (int off = DP >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, DP);
    return v;
}

// LDS hand-off between lanes of ONE wavefront: DS ops of a wave execute in
// issue order, so only the compiler has to be told not to reorder.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#ifndef SPRINTZ_THREADS
#define SPRINTZ_THREADS 256
#endif
constexpr int kThreads = SPRINTZ_THREADS;        // wavefronts per workgroup x 64

// The verbatim tail of a stream (:1171) -- for chunks shorter than one group, the whole chunk (BASELINE config 3 at
// 1 KB): `nlanes` lanes copy nbytes from t to d.  16 bytes per lane per trip, FOUR trips' loads issued before the first
// store (t and d may not alias, but the compiler cannot know: written as load-store pairs, every trip waited for a
// round trip to memory -- 0.352 ms for 512 MiB of 1 KB chunks); NT: written non-temporally (decoded samples); the tail bytewise.
template <bool NT = true>
__device__ __forceinline__ void copy_verbatim(const uint8_t* __restrict__ t, uint8_t* __restrict__ d, uint32_t nbytes, uint32_t lane, uint32_t nlanes)
{
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    typedef v4u __attribute__((aligned(1), may_alias)) v4u_a1;
    const uint32_t n16 = nbytes >> 4;
    for (uint32_t j0 = lane; j0 < n16; j0 += 4u * nlanes) {
        v4u v[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            const uint32_t j = j0 + k * nlanes;
            if (j < n16) v[k] = *(const v4u_a1*)(t + 16u * j);
        }
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            const uint32_t j = j0 + k * nlanes;
            if (j < n16) {
                if constexpr (NT) __builtin_nontemporal_store(v[k], (v4u_a1*)(d + 16u * j));
                else *(v4u_a1*)(d + 16u * j) = v[k];
            }
        }
    }
    for (uint32_t j = (n16 << 4) + lane; j < nbytes; j += nlanes) d[j] = t[j];
}


}  // namespace sprintz

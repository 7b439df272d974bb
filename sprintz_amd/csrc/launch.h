// launch.h -- host-side dispatch onto the template instantiations.  Each
// (width, direction) pair lives in its own translation unit so that hipcc can
// build them in parallel.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>

#include "lds_attr.h"
#include "decode_kernel.h"
#include "decode_fast.h"
#include "decode_uni.h"
#include "encode_kernel.h"
#include "encode_fast.h"
#include "encode_uni.h"
#include "encode_wide.h"

namespace sprintz {

// columns-per-lane values that are instantiated (general layout); low-dim uses CPL = 1
constexpr int kCplSet[] = {1, 2, 3, 4, 5, 6, 8};

hipError_t launch_decode_w8(bool fire, bool lowdim, int cpl, int q, unsigned grid, size_t shmem, hipStream_t st, const DecodeArgs& a);
hipError_t launch_decode_w16(bool fire, bool lowdim, int cpl, int q, unsigned grid, size_t shmem, hipStream_t st, const DecodeArgs& a);
// fast path: general layout, one column per lane, LDS-transposed stores (see decode_fast.h)
// (ds: columns the LDS carve is sized for when that is fewer than dp * cpl -- decode_fast.h, DS; 0 = dp * cpl)
hipError_t launch_decode_fast_w8(bool fire, int dp, int cpl, bool exact, int q, int ds, unsigned grid, size_t shmem, hipStream_t st, const DecodeArgs& a);
hipError_t launch_decode_fast_w16(bool fire, int dp, int cpl, bool exact, int q, int ds, unsigned grid, size_t shmem, hipStream_t st, const DecodeArgs& a);
// small batches: one workgroup per chunk, both layouts, 1 .. 64 columns (decode_lat.h); bound_bytes = the longest stream a chunk can have
hipError_t launch_decode_lat(int w, bool fire, int dp, bool lowdim, unsigned grid, uint32_t bound_bytes, hipStream_t st, const DecodeArgs& a);
hipError_t launch_encode_lat(int w, bool fire, int dp, bool lowdim, unsigned grid, uint32_t bound_bytes, hipStream_t st, const EncodeArgs& a);   // encode_lat.h
// streams of 513 .. 2047 columns: one workgroup per chunk, <= 8 columns per lane (any_ndims.hip); shmem = the encoder's group window
hipError_t launch_decode_any(int w, bool fire, unsigned grid, hipStream_t st, const DecodeArgs& a);
hipError_t launch_encode_any(int w, bool fire, unsigned grid, size_t shmem, hipStream_t st, const EncodeArgs& a);
// streams of 2 048 .. 65 535 columns: the same scheme in column tiles, per-column state behind the rows / in `counters` (nchunks x ndims int32; FIRE only)
hipError_t launch_decode_big(int w, bool fire, unsigned grid, hipStream_t st, const DecodeArgs& a, int32_t* counters);
hipError_t launch_encode_big(int w, bool fire, unsigned grid, hipStream_t st, const EncodeArgs& a, int32_t* counters);
// low-dim streams with 1, 2 or 4 columns (8 bits) / 1 or 2 (16 bits), one lane per chunk (decode_uni.h)
hipError_t launch_decode_uni_w8(bool fire, int nd, int q, unsigned grid, hipStream_t st, const DecodeArgs& a);
hipError_t launch_decode_uni_w16(bool fire, int nd, int q, unsigned grid, hipStream_t st, const DecodeArgs& a);
// streams of 65 .. 128 columns, two columns per lane (encode_wide.h)
hipError_t launch_encode_wide_w8(bool fire, bool exact, unsigned grid, size_t shmem, hipStream_t st, const EncodeArgs& a);
hipError_t launch_encode_wide_w16(bool fire, bool exact, unsigned grid, size_t shmem, hipStream_t st, const EncodeArgs& a);
// two columns per lane for streams of up to 64 columns: dp = 4 / 8 / 16 / 32 lanes a chunk (encode_wide.h, DPT)
hipError_t launch_encode_pair_w8(bool fire, int dp, bool exact, unsigned grid, size_t shmem, hipStream_t st, const EncodeArgs& a);
hipError_t launch_encode_pair_w16(bool fire, int dp, bool exact, unsigned grid, size_t shmem, hipStream_t st, const EncodeArgs& a);
// 8-bit streams of 65 .. 80 columns on 32 lanes a chunk (encode_wide.h, SPLIT)
hipError_t launch_encode_split_w8(bool fire, unsigned grid, size_t shmem, hipStream_t st, const EncodeArgs& a);
hipError_t launch_encode_uni_w8(bool fire, int nd, unsigned grid, hipStream_t st, const EncodeArgs& a);
hipError_t launch_encode_uni_w16(bool fire, int nd, unsigned grid, hipStream_t st, const EncodeArgs& a);
hipError_t launch_encode_fast_w8(bool fire, int dp, bool exact, unsigned grid, size_t shmem, hipStream_t st, const EncodeArgs& a);
hipError_t launch_encode_fast_w16(bool fire, int dp, bool exact, unsigned grid, size_t shmem, hipStream_t st, const EncodeArgs& a);
hipError_t launch_encode_w8(bool fire, bool lowdim, int cpl, unsigned grid, size_t shmem, hipStream_t st, const EncodeArgs& a);
hipError_t launch_encode_w16(bool fire, bool lowdim, int cpl, unsigned grid, size_t shmem, hipStream_t st, const EncodeArgs& a);

// records `what` as this thread's last error (sprintz_mi355x_last_error) and returns `code`: every
// failing return of every translation unit goes through it, so the message is never stale
int set_error(int code, const char* what);

// the calling thread's pooled scratch for host-pointer entry points (api.hip: acquire_scratch): buffers only grow, the
// stream is private and non-blocking; valid until the thread's next host_scratch() call
struct HostScratch { hipStream_t stream; uint8_t* dev; uint8_t* pin; };
int host_scratch(size_t dev_bytes, size_t pin_bytes, HostScratch* out);
bool have_device();                       // probed once per process
std::atomic<long long>& huf0_sync_chunks(); // huf0.hip: batch size up to which the stream stage runs as a wave per chunk with self-synchronising decoders (huf0_sync.h)
std::atomic<long long>& huf0_big_batch(); // huf0.hip: batch size from which the one-table stream kernel runs as workgroups of HUF0_BIG_WG (2) waves

hipError_t launch_size_scan(const uint32_t* d_sizes, uint64_t n, uint32_t align, uint64_t* d_offsets, void* d_tmp, hipStream_t st);

template <typename K, typename A>
inline hipError_t launch_one(K kernel, unsigned grid, size_t shmem, hipStream_t st, const A& a)
{
    if (shmem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kThreads), shmem, st, a);
    return hipGetLastError();
}

#define SPRINTZ_DISPATCH_ENC(KERNEL, W)                                                               \
    if (lowdim) {                                                                                     \
        if (cpl != 1) return hipErrorInvalidValue;                                                    \
        return fire ? launch_one(KERNEL<W, true, true, 1>, grid, shmem, st, a)                        \
                    : launch_one(KERNEL<W, false, true, 1>, grid, shmem, st, a);                      \
    }                                                                                                 \
    switch (cpl) {                                                                                    \
        case 1: return fire ? launch_one(KERNEL<W, true, false, 1>, grid, shmem, st, a)               \
                            : launch_one(KERNEL<W, false, false, 1>, grid, shmem, st, a);             \
        case 2: return fire ? launch_one(KERNEL<W, true, false, 2>, grid, shmem, st, a)               \
                            : launch_one(KERNEL<W, false, false, 2>, grid, shmem, st, a);             \
        case 3: return fire ? launch_one(KERNEL<W, true, false, 3>, grid, shmem, st, a)               \
                            : launch_one(KERNEL<W, false, false, 3>, grid, shmem, st, a);             \
        case 4: return fire ? launch_one(KERNEL<W, true, false, 4>, grid, shmem, st, a)               \
                            : launch_one(KERNEL<W, false, false, 4>, grid, shmem, st, a);             \
        case 5: return fire ? launch_one(KERNEL<W, true, false, 5>, grid, shmem, st, a)               \
                            : launch_one(KERNEL<W, false, false, 5>, grid, shmem, st, a);             \
        case 6: return fire ? launch_one(KERNEL<W, true, false, 6>, grid, shmem, st, a)               \
                            : launch_one(KERNEL<W, false, false, 6>, grid, shmem, st, a);             \
        case 8: return fire ? launch_one(KERNEL<W, true, false, 8>, grid, shmem, st, a)               \
                            : launch_one(KERNEL<W, false, false, 8>, grid, shmem, st, a);             \
        default: return hipErrorInvalidValue;                                                         \
    }

#define SPRINTZ_DISPATCH_Q(KERNEL, W, Q)                                                              \
    if (lowdim) {                                                                                     \
        if (cpl != 1) return hipErrorInvalidValue;                                                    \
        return fire ? launch_one(KERNEL<W, true, true, 1, Q>, grid, shmem, st, a)                     \
                    : launch_one(KERNEL<W, false, true, 1, Q>, grid, shmem, st, a);                   \
    }                                                                                                 \
    switch (cpl) {                                                                                    \
        case 1: return fire ? launch_one(KERNEL<W, true, false, 1, Q>, grid, shmem, st, a)            \
                            : launch_one(KERNEL<W, false, false, 1, Q>, grid, shmem, st, a);          \
        case 2: return fire ? launch_one(KERNEL<W, true, false, 2, Q>, grid, shmem, st, a)            \
                            : launch_one(KERNEL<W, false, false, 2, Q>, grid, shmem, st, a);          \
        case 3: return fire ? launch_one(KERNEL<W, true, false, 3, Q>, grid, shmem, st, a)            \
                            : launch_one(KERNEL<W, false, false, 3, Q>, grid, shmem, st, a);          \
        case 4: return fire ? launch_one(KERNEL<W, true, false, 4, Q>, grid, shmem, st, a)            \
                            : launch_one(KERNEL<W, false, false, 4, Q>, grid, shmem, st, a);          \
        case 5: return fire ? launch_one(KERNEL<W, true, false, 5, Q>, grid, shmem, st, a)            \
                            : launch_one(KERNEL<W, false, false, 5, Q>, grid, shmem, st, a);          \
        case 6: return fire ? launch_one(KERNEL<W, true, false, 6, Q>, grid, shmem, st, a)            \
                            : launch_one(KERNEL<W, false, false, 6, Q>, grid, shmem, st, a);          \
        case 8: return fire ? launch_one(KERNEL<W, true, false, 8, Q>, grid, shmem, st, a)            \
                            : launch_one(KERNEL<W, false, false, 8, Q>, grid, shmem, st, a);          \
        default: return hipErrorInvalidValue;                                                         \
    }

// q: kQueryOff / kQueryMaterialize / kQueryReduceOnly (decode_kernel.h)
#define SPRINTZ_DISPATCH(KERNEL, W)                                                                   \
    if (q == kQueryOff) { SPRINTZ_DISPATCH_Q(KERNEL, W, kQueryOff) }                                  \
    if (q == kQueryMaterialize) { SPRINTZ_DISPATCH_Q(KERNEL, W, kQueryMaterialize) }                  \
    SPRINTZ_DISPATCH_Q(KERNEL, W, kQueryReduceOnly)

#define SPRINTZ_FAST_CASE(KERNEL, W, DPV, CPLV, Q, CMV)                                              \
    if (dp == DPV && cpl == CPLV) {                                                                   \
        if (exact) return fire ? launch_one(KERNEL<W, true, DPV, CPLV, true, Q, CMV>, grid, shmem, st, a)  \
                               : launch_one(KERNEL<W, false, DPV, CPLV, true, Q, CMV>, grid, shmem, st, a); \
        return fire ? launch_one(KERNEL<W, true, DPV, CPLV, false, Q, CMV>, grid, shmem, st, a)       \
                    : launch_one(KERNEL<W, false, DPV, CPLV, false, Q, CMV>, grid, shmem, st, a);     \
    }

// decoder fast path: one column per lane for D <= 64, 2 / 4 columns per lane of a
// 64-lane group for D <= 128 / 256
#define SPRINTZ_DISPATCH_DECODE_FAST_Q(KERNEL, W, Q, CMV)                                             \
    SPRINTZ_FAST_CASE(KERNEL, W, 4, 1, Q, CMV)                                                        \
    SPRINTZ_FAST_CASE(KERNEL, W, 8, 1, Q, CMV)                                                        \
    SPRINTZ_FAST_CASE(KERNEL, W, 16, 1, Q, CMV)                                                       \
    SPRINTZ_FAST_CASE(KERNEL, W, 32, 1, Q, CMV)                                                       \
    SPRINTZ_FAST_CASE(KERNEL, W, 64, 1, Q, CMV)                                                       \
    SPRINTZ_FAST_CASE(KERNEL, W, 64, 2, Q, CMV)                                                       \
    SPRINTZ_FAST_CASE(KERNEL, W, 64, 4, Q, CMV)                                                       \
    return hipErrorInvalidValue;

// the column-major destination exists for the plain decode only (a reduce-only query writes
// no output, so its layout does not matter)
#define SPRINTZ_DISPATCH_DECODE_FAST(KERNEL, W)                                                       \
    if (q == kQueryOff && a.col_stride) { SPRINTZ_DISPATCH_DECODE_FAST_Q(KERNEL, W, kQueryOff, true) } \
    if (a.col_stride && q == kQueryMaterialize) return hipErrorInvalidValue;                          \
    if (q == kQueryOff) { SPRINTZ_DISPATCH_DECODE_FAST_Q(KERNEL, W, kQueryOff, false) }               \
    if (q == kQueryMaterialize) { SPRINTZ_DISPATCH_DECODE_FAST_Q(KERNEL, W, kQueryMaterialize, false) } \
    SPRINTZ_DISPATCH_DECODE_FAST_Q(KERNEL, W, kQueryReduceOnly, false)

#define SPRINTZ_ENC_FAST_CASE(KERNEL, W, DPV, CMV)                                                   \
    case DPV:                                                                                         \
        if (exact) return fire ? launch_one(KERNEL<W, true, DPV, true, CMV>, grid, shmem, st, a)      \
                               : launch_one(KERNEL<W, false, DPV, true, CMV>, grid, shmem, st, a);    \
        return fire ? launch_one(KERNEL<W, true, DPV, false, CMV>, grid, shmem, st, a)                \
                    : launch_one(KERNEL<W, false, DPV, false, CMV>, grid, shmem, st, a);

#define SPRINTZ_DISPATCH_FAST_CM(KERNEL, W, CMV)                                                      \
    switch (dp) {                                                                                     \
        SPRINTZ_ENC_FAST_CASE(KERNEL, W, 4, CMV)                                                      \
        SPRINTZ_ENC_FAST_CASE(KERNEL, W, 8, CMV)                                                      \
        SPRINTZ_ENC_FAST_CASE(KERNEL, W, 16, CMV)                                                     \
        SPRINTZ_ENC_FAST_CASE(KERNEL, W, 32, CMV)                                                     \
        SPRINTZ_ENC_FAST_CASE(KERNEL, W, 64, CMV)                                                     \
        default: return hipErrorInvalidValue;                                                         \
    }

// a.col_stride != 0 selects the column-major source variant
#define SPRINTZ_DISPATCH_FAST(KERNEL, W)                                                              \
    if (a.col_stride) { SPRINTZ_DISPATCH_FAST_CM(KERNEL, W, true) }                                   \
    SPRINTZ_DISPATCH_FAST_CM(KERNEL, W, false)

}  // namespace sprintz

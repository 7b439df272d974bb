// transforms.hip -- the reference's stand-alone transforms on the GPU:
//   kind 0  delta         encode/decode_delta_rowmajor        cpp/Compress/delta.cpp:35-121, :133-397
//   kind 1  double delta  encode/decode_doubledelta_rowmajor  delta.cpp:405-529, :532-693
// Per column c (element index mod ndims), state starting at zero, arithmetic wrapping at the
// element width:   delta  y[r] = x[r] - x[r-1];   double delta  y[r] = x[r] - 2 x[r-1] + x[r-2].
//
// ENCODE is element-wise.  DECODE is a per-column recurrence over the rows of ONE long stream
// (the reference decodes it serially), so it is a scan: with the state (x, d) entering a run
// of n rows, the state leaving it is
//     d' = d + S1,   x' = x + n d + S2,      S1 = sum of y,  S2 = sum of the running sums of y
// (delta is the case d == 0, S2 == S1), and two consecutive runs combine associatively:
//     (S1, S2, n) o (S1', S2', n') = (S1 + S1', S2 + n' S1 + S2', n + n').
// Level k of the scan holds one summary per R^k rows; `reduce` builds level k+1 from level k,
// `apply` walks back down handing every run its incoming state, and at level 0 writes x.
// Everything is modulo 2^W, so the summaries are stored in the element type.
#include "../../include/sprintz_mi355x.h"

#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

namespace {

constexpr int R = 16;            // rows (or lower-level runs) folded by one thread
constexpr int kTB = 256;

template <typename U> struct Level {
    const U* s1;                 // level 0: the input y itself
    const U* s2;
    U* xin;                      // incoming state of every run of this level (levels >= 1)
    U* din;
    uint64_t rows;               // runs at this level
    uint64_t span;               // original rows per run (R^k)
};

// ---------------------------------------------------------------- encode: y = x - x[-D] (- ...)
template <typename U, int KIND>
__global__ void __launch_bounds__(kTB) encode_kernel(const U* x, uint64_t len, uint32_t D, U* y)
{
    constexpr int V = 16 / sizeof(U);
    const uint64_t i0 = ((uint64_t)blockIdx.x * kTB + threadIdx.x) * V;
    if (i0 >= len) return;
    U a[V], p1[V], p2[V];
    const bool full = i0 + V <= len;
    if (full && i0 >= 2ull * D) {                      // the common case: three 16-byte loads, one store
        typedef uint32_t v4 __attribute__((ext_vector_type(4)));
        typedef v4 __attribute__((aligned(1), may_alias)) v4u;
        *(v4*)a = *(const v4u*)(x + i0);
        *(v4*)p1 = *(const v4u*)(x + i0 - D);
        if (KIND) *(v4*)p2 = *(const v4u*)(x + i0 - 2ull * D);
        U o[V];
#pragma unroll
        for (int k = 0; k < V; k++) o[k] = KIND ? (U)(a[k] - 2 * p1[k] + p2[k]) : (U)(a[k] - p1[k]);
        *(v4u*)(y + i0) = *(v4*)o;
        return;
    }
    for (int k = 0; k < V && i0 + k < len; k++) {
        const uint64_t i = i0 + k;
        const U q1 = i >= D ? x[i - D] : (U)0, q2 = i >= 2ull * D ? x[i - 2ull * D] : (U)0;
        y[i] = KIND ? (U)(x[i] - 2 * q1 + q2) : (U)(x[i] - q1);
    }
}

// ---------------------------------------------------------------- decode, up: level k -> k+1
// thread = (run s of level k+1, column c); c runs fastest so that a wavefront reads whole rows
template <typename U, int KIND>
__global__ void __launch_bounds__(kTB) reduce_kernel(Level<U> lo, uint64_t len, uint32_t D, uint64_t rows0, U* s1_out, U* s2_out,
                                                     uint64_t rows_out)
{
    const uint64_t tid = (uint64_t)blockIdx.x * kTB + threadIdx.x;
    if (tid >= rows_out * D) return;
    const uint64_t s = tid / D;
    const uint32_t c = (uint32_t)(tid - s * D);
    U S1 = 0, S2 = 0;
    for (int j = 0; j < R; j++) {
        const uint64_t r = s * R + j;
        if (r >= lo.rows) break;
        const uint64_t e = r * D + c;
        U c1, c2, n;
        if (lo.span == 1) {                            // a row of the input: (y, y, 1); a missing element of a ragged last row: 0
            c1 = e < len ? lo.s1[e] : (U)0;
            c2 = c1;
            n = 1;
        } else {
            c1 = lo.s1[e];
            c2 = KIND ? lo.s2[e] : c1;
            const uint64_t first = r * lo.span;
            n = (U)(rows0 - first < lo.span ? rows0 - first : lo.span);
        }
        if (KIND) S2 = (U)(S2 + (U)(n * S1) + c2);     // S2 of the run so far + n' S1 + S2'
        S1 = (U)(S1 + c1);
    }
    s1_out[tid] = S1;
    if (KIND) s2_out[tid] = S2;
}

// ---------------------------------------------------------------- decode, down: hand every run of level k its incoming state
template <typename U, int KIND>
__global__ void __launch_bounds__(kTB) apply_kernel(Level<U> lo, uint64_t len, uint32_t D, uint64_t rows0, const U* xin_hi, const U* din_hi,
                                                    uint64_t rows_hi, U* dest)
{
    const uint64_t tid = (uint64_t)blockIdx.x * kTB + threadIdx.x;
    if (tid >= rows_hi * D) return;
    const uint64_t s = tid / D;
    const uint32_t c = (uint32_t)(tid - s * D);
    U x = xin_hi ? xin_hi[tid] : (U)0;                 // the top level starts from the zero state
    U d = (KIND && din_hi) ? din_hi[tid] : (U)0;
    for (int j = 0; j < R; j++) {
        const uint64_t r = s * R + j;
        if (r >= lo.rows) break;
        const uint64_t e = r * D + c;
        if (lo.span == 1) {
            if (e >= len) break;
            const U y = lo.s1[e];
            if (KIND) { d = (U)(d + y); x = (U)(x + d); }
            else x = (U)(x + y);
            dest[e] = x;
        } else {
            lo.xin[e] = x;
            if (KIND) lo.din[e] = d;
            const U c1 = lo.s1[e];
            if (KIND) {
                const uint64_t first = r * lo.span;
                const U n = (U)(rows0 - first < lo.span ? rows0 - first : lo.span);
                x = (U)(x + (U)(n * d) + lo.s2[e]);
                d = (U)(d + c1);
            } else {
                x = (U)(x + c1);
            }
        }
    }
}

thread_local std::string g_err;
int fail(int code, const char* what)
{
    g_err = what;
    return code;
}

struct Plan {
    std::vector<uint64_t> rows;      // rows[k] = runs at level k (rows[0] = input rows)
    std::vector<uint64_t> span;
    size_t tmp_bytes = 0;
};

Plan make_plan(int kind, int esz, uint64_t len, uint32_t D)
{
    Plan p;
    uint64_t rows = (len + D - 1) / D, span = 1;
    p.rows.push_back(rows);
    p.span.push_back(span);
    while (rows > 1) {                                   // the top level is a single run
        rows = (rows + R - 1) / R;
        span *= R;
        p.rows.push_back(rows);
        p.span.push_back(span);
    }
    const size_t arrays = kind ? 4 : 2;                  // s1 [, s2], xin [, din] per level >= 1
    for (size_t k = 1; k < p.rows.size(); k++) p.tmp_bytes += ((p.rows[k] * D * (size_t)esz + 63) & ~(size_t)63) * arrays;
    return p;
}

template <typename U, int KIND>
int decode_device(const U* y, uint64_t len, uint32_t D, U* dest, uint8_t* tmp, hipStream_t st)
{
    const Plan p = make_plan(KIND, sizeof(U), len, D);
    const size_t L = p.rows.size();
    std::vector<Level<U>> lv(L);
    lv[0] = Level<U>{y, y, nullptr, nullptr, p.rows[0], 1};
    uint8_t* t = tmp;
    auto take = [&](uint64_t n) { U* q = (U*)t; t += (n * D * sizeof(U) + 63) & ~(size_t)63; return q; };
    for (size_t k = 1; k < L; k++) {
        U* s1 = take(p.rows[k]);
        U* s2 = KIND ? take(p.rows[k]) : nullptr;
        U* xi = take(p.rows[k]);
        U* di = KIND ? take(p.rows[k]) : nullptr;
        lv[k] = Level<U>{s1, s2, xi, di, p.rows[k], p.span[k]};
    }
    const uint64_t rows0 = p.rows[0];
    for (size_t k = 0; k + 1 < L; k++) {
        const uint64_t threads = lv[k + 1].rows * D;
        hipLaunchKernelGGL((reduce_kernel<U, KIND>), dim3((unsigned)((threads + kTB - 1) / kTB)), dim3(kTB), 0, st, lv[k], len, D, rows0,
                           (U*)lv[k + 1].s1, (U*)lv[k + 1].s2, lv[k + 1].rows);
    }
    if (L == 1) {                                        // a single row: one pseudo-run above it
        hipLaunchKernelGGL((apply_kernel<U, KIND>), dim3((unsigned)((D + kTB - 1) / kTB)), dim3(kTB), 0, st, lv[0], len, D, rows0,
                           (const U*)nullptr, (const U*)nullptr, (uint64_t)1, dest);
    }
    for (size_t k = L - 1; k >= 1; k--) {
        const bool top = k == L - 1;
        const uint64_t threads = lv[k].rows * D;
        hipLaunchKernelGGL((apply_kernel<U, KIND>), dim3((unsigned)((threads + kTB - 1) / kTB)), dim3(kTB), 0, st, lv[k - 1], len, D, rows0,
                           top ? (const U*)nullptr : (const U*)lv[k].xin, top ? (const U*)nullptr : (const U*)lv[k].din, lv[k].rows, dest);
    }
    return hipGetLastError() == hipSuccess ? 0 : fail(SPRINTZ_E_HIP, "transform decode launch");
}

template <typename U, int KIND>
int encode_device(const U* x, uint64_t len, uint32_t D, U* y, hipStream_t st)
{
    constexpr int V = 16 / sizeof(U);
    const uint64_t threads = (len + V - 1) / V;
    hipLaunchKernelGGL((encode_kernel<U, KIND>), dim3((unsigned)((threads + kTB - 1) / kTB)), dim3(kTB), 0, st, x, len, D, y);
    return hipGetLastError() == hipSuccess ? 0 : fail(SPRINTZ_E_HIP, "transform encode launch");
}

int check(int kind, int esz, uint64_t len, uint16_t ndims)
{
    if (kind != SPRINTZ_TRANSFORM_DELTA && kind != SPRINTZ_TRANSFORM_DOUBLEDELTA) return fail(SPRINTZ_E_INVALID, "kind must be 0 (delta) or 1 (double delta)");
    if (esz != 1 && esz != 2) return fail(SPRINTZ_E_INVALID, "elem_bytes must be 1 or 2");
    if (ndims == 0) return fail(SPRINTZ_E_INVALID, "ndims == 0");
    if (len / 64 > 0x7fffffffull) return fail(SPRINTZ_E_INVALID, "len too large for one launch");
    return 0;
}

bool have_device()
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess && n > 0;
}

}  // namespace

extern "C" {

const char* sprintz_mi355x_transform_last_error(void) { return g_err.c_str(); }

size_t sprintz_mi355x_transform_tmp_bytes(int kind, int elem_bytes, uint64_t len, uint16_t ndims)
{
    if (ndims == 0 || (elem_bytes != 1 && elem_bytes != 2)) return 0;
    return make_plan(kind, elem_bytes, len, ndims).tmp_bytes + 64;
}

int sprintz_mi355x_transform_encode_device(int kind, int elem_bytes, const void* d_src, uint64_t len, uint16_t ndims, void* d_dest,
                                           void* hip_stream)
{
    int rc = check(kind, elem_bytes, len, ndims);
    if (rc) return rc;
    if (!d_src || !d_dest) return fail(SPRINTZ_E_INVALID, "null device pointer");
    if (!have_device()) return fail(SPRINTZ_E_NO_DEVICE, "no HIP device; there is no CPU fallback");
    if (len == 0) return 0;
    hipStream_t st = (hipStream_t)hip_stream;
    if (elem_bytes == 1) return kind ? encode_device<uint8_t, 1>((const uint8_t*)d_src, len, ndims, (uint8_t*)d_dest, st)
                                     : encode_device<uint8_t, 0>((const uint8_t*)d_src, len, ndims, (uint8_t*)d_dest, st);
    return kind ? encode_device<uint16_t, 1>((const uint16_t*)d_src, len, ndims, (uint16_t*)d_dest, st)
                : encode_device<uint16_t, 0>((const uint16_t*)d_src, len, ndims, (uint16_t*)d_dest, st);
}

int sprintz_mi355x_transform_decode_device(int kind, int elem_bytes, const void* d_src, uint64_t len, uint16_t ndims, void* d_dest,
                                           void* d_tmp, void* hip_stream)
{
    int rc = check(kind, elem_bytes, len, ndims);
    if (rc) return rc;
    if (!d_src || !d_dest || !d_tmp) return fail(SPRINTZ_E_INVALID, "null device pointer");
    if (!have_device()) return fail(SPRINTZ_E_NO_DEVICE, "no HIP device; there is no CPU fallback");
    if (len == 0) return 0;
    hipStream_t st = (hipStream_t)hip_stream;
    if (elem_bytes == 1) return kind ? decode_device<uint8_t, 1>((const uint8_t*)d_src, len, ndims, (uint8_t*)d_dest, (uint8_t*)d_tmp, st)
                                     : decode_device<uint8_t, 0>((const uint8_t*)d_src, len, ndims, (uint8_t*)d_dest, (uint8_t*)d_tmp, st);
    return kind ? decode_device<uint16_t, 1>((const uint16_t*)d_src, len, ndims, (uint16_t*)d_dest, (uint8_t*)d_tmp, st)
                : decode_device<uint16_t, 0>((const uint16_t*)d_src, len, ndims, (uint16_t*)d_dest, (uint8_t*)d_tmp, st);
}

// host forms with the reference's container: 6-byte header {u32 len; u16 ndims} (format.h:65-86)
int64_t sprintz_mi355x_transform_encode(int kind, int elem_bytes, const void* src, uint32_t len, void* dest, uint16_t ndims, int write_size)
{
    if (kind != 0 && kind != 1) return fail(SPRINTZ_E_INVALID, "kind must be 0 (delta) or 1 (double delta)");
    if (elem_bytes != 1 && elem_bytes != 2) return fail(SPRINTZ_E_INVALID, "elem_bytes must be 1 or 2");
    if (!src || !dest) return fail(SPRINTZ_E_INVALID, "null pointer");
    if (!have_device()) return fail(SPRINTZ_E_NO_DEVICE, "no HIP device; there is no CPU fallback");
    uint8_t* d = (uint8_t*)dest;
    int64_t hdr = 0;
    if (write_size) {
        memcpy(d, &len, 4);
        memcpy(d + 4, &ndims, 2);
        hdr = elem_bytes == 1 ? 6 : 3;
        d += 6;
    }
    if (len == 0 || ndims == 0) return (int64_t)len + hdr;
    void *dx = nullptr, *dy = nullptr;
    const size_t nb = (size_t)len * elem_bytes;
    if (hipMalloc(&dx, nb) != hipSuccess || hipMalloc(&dy, nb) != hipSuccess) { (void)hipFree(dx); return fail(SPRINTZ_E_HIP, "hipMalloc"); }
    int rc = hipMemcpy(dx, src, nb, hipMemcpyHostToDevice) == hipSuccess ? 0 : SPRINTZ_E_HIP;
    if (!rc) rc = sprintz_mi355x_transform_encode_device(kind, elem_bytes, dx, len, ndims, dy, nullptr);
    if (!rc && hipMemcpy(d, dy, nb, hipMemcpyDeviceToHost) != hipSuccess) rc = SPRINTZ_E_HIP;
    (void)hipFree(dx);
    (void)hipFree(dy);
    return rc ? rc : (int64_t)len + hdr;
}

// src carries the header unless len/ndims are given (the reference's 4-argument decode form, delta.h:19-21)
int64_t sprintz_mi355x_transform_decode(int kind, int elem_bytes, const void* src, void* dest, uint32_t raw_len, uint16_t raw_ndims)
{
    if (kind != 0 && kind != 1) return fail(SPRINTZ_E_INVALID, "kind must be 0 (delta) or 1 (double delta)");
    if (elem_bytes != 1 && elem_bytes != 2) return fail(SPRINTZ_E_INVALID, "elem_bytes must be 1 or 2");
    if (!src || !dest) return fail(SPRINTZ_E_INVALID, "null pointer");
    if (!have_device()) return fail(SPRINTZ_E_NO_DEVICE, "no HIP device; there is no CPU fallback");
    const uint8_t* s = (const uint8_t*)src;
    uint32_t len = raw_len;
    uint16_t ndims = raw_ndims;
    if (raw_ndims == 0 && raw_len == 0) {
        memcpy(&len, s, 4);
        memcpy(&ndims, s + 4, 2);
        s += 6;
    }
    if (ndims == 0) return 0;                                  // delta.cpp:191, :637
    if (len == 0) return 0;
    const size_t nb = (size_t)len * elem_bytes;
    const size_t tb = sprintz_mi355x_transform_tmp_bytes(kind, elem_bytes, len, ndims);
    void *dy = nullptr, *dx = nullptr, *dt = nullptr;
    if (hipMalloc(&dy, nb) != hipSuccess || hipMalloc(&dx, nb) != hipSuccess || hipMalloc(&dt, tb) != hipSuccess) {
        (void)hipFree(dy); (void)hipFree(dx);
        return fail(SPRINTZ_E_HIP, "hipMalloc");
    }
    int rc = hipMemcpy(dy, s, nb, hipMemcpyHostToDevice) == hipSuccess ? 0 : SPRINTZ_E_HIP;
    if (!rc) rc = sprintz_mi355x_transform_decode_device(kind, elem_bytes, dy, len, ndims, dx, dt, nullptr);
    if (!rc && hipMemcpy(dest, dx, nb, hipMemcpyDeviceToHost) != hipSuccess) rc = SPRINTZ_E_HIP;
    (void)hipFree(dy);
    (void)hipFree(dx);
    (void)hipFree(dt);
    return rc ? rc : (int64_t)len;
}

}  // extern "C"

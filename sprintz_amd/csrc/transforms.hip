// transforms.hip -- the reference's stand-alone transforms on the GPU:
//   kind 0  delta         encode/decode_delta_rowmajor        cpp/Compress/delta.cpp:35-121, :133-397
//   kind 1  double delta  encode/decode_doubledelta_rowmajor  delta.cpp:405-529, :532-693
//   kind 2  xff (FIRE)    encode/decode_xff_rowmajor          cpp/Compress/predict.cpp:57-289, :302-517
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
//
// XFF (the FIRE forecaster without packing, with predict.cpp's own constants) is different in
// kind: the counter a block leaves depends non-linearly (>> truncation, sign()) on the counter
// it entered with, so ONE stream offers no parallelism along the rows in either direction --
// only across its columns.  xff_kernel is therefore a lane per column walking the rows, the
// next block's samples in flight while one is forecast; it is latency-bound by construction
// ("replicas only") and is here so that the whole transform surface of the
// reference exists behind one boundary.  Many independent series belong in the batched codec.
#include "../../include/sprintz_mi355x.h"

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
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

// ---------------------------------------------------------------- decode: the runs' incoming states in ONE launch
// The wave-scan path leaves one summary per run (a few thousand of them for a 128 MiB stream); handing every run its incoming state
// through the generic levels is three reduce and three apply launches of ~4.6 us each -- 28 of the 119 us of a 64 Mi-sample decode
// (rocprofv3 kernel trace).  One workgroup does it: thread (g, c) composes the runs of its slice g of column c, the slices' compositions
// meet in LDS, every thread folds the slices before its own and walks its runs again writing their incoming states.
// Same arithmetic as reduce_kernel / apply_kernel (composition (S1, S2, n) o (S1', S2', n') = (S1 + S1', S2 + n' S1 + S2', n + n')).
constexpr int kTopT = 1024;
template <typename U, int KIND>
__global__ void __launch_bounds__(kTopT) scan_runs_kernel(Level<U> lo, uint32_t D, uint64_t rows0, uint32_t G)
{
    __shared__ U sm1[kTopT], sm2[kTopT], smn[kTopT];
    const uint32_t t = threadIdx.x, g = t / D, c = t - g * D;
    const bool on = g < G;
    const uint64_t per = (lo.rows + G - 1) / G, r0 = (uint64_t)g * per, r1 = r0 + per < lo.rows ? r0 + per : lo.rows;
    auto rows_of = [&](uint64_t r) -> U { const uint64_t first = r * lo.span; return (U)(rows0 - first < lo.span ? rows0 - first : lo.span); };
    U S1 = 0, S2 = 0, N = 0;
    if (on)
        for (uint64_t r = r0; r < r1; r++) {
            const uint64_t e = r * D + c;
            const U c1 = lo.s1[e], n = rows_of(r);
            if (KIND) S2 = (U)(S2 + (U)(n * S1) + lo.s2[e]);
            S1 = (U)(S1 + c1);
            N = (U)(N + n);
        }
    if (on) { sm1[t] = S1; sm2[t] = S2; smn[t] = N; }
    __syncthreads();
    if (!on) return;
    U x = 0, d = 0;                                    // the state before slice g: the slices before it applied to the zero state
    for (uint32_t k = 0; k < g; k++) {
        const uint32_t i = k * D + c;
        if (KIND) { x = (U)(x + (U)(smn[i] * d) + sm2[i]); d = (U)(d + sm1[i]); }
        else x = (U)(x + sm1[i]);
    }
    for (uint64_t r = r0; r < r1; r++) {
        const uint64_t e = r * D + c;
        lo.xin[e] = x;
        if (KIND) {
            lo.din[e] = d;
            x = (U)(x + (U)(rows_of(r) * d) + lo.s2[e]);
            d = (U)(d + lo.s1[e]);
        } else {
            x = (U)(x + lo.s1[e]);
        }
    }
}

// ---------------------------------------------------------------- decode, small rows: one wavefront scans a run of rows
// When a row is a whole number (<= 64) of 16-byte / 4-byte / element-sized pieces, the
// lanes of a wavefront share a contiguous 4 KB / 1 KB / 256-element load: a lane folds
// kRowsPerLane consecutive rows of its column piece in registers, the lane summaries are scanned
// across lanes at stride Dv with the composition above (the right operand of step s covers exactly
// kRowsPerLane * 2^s rows), every lane advances the carried state over the rows before its own,
// finishes its rows, stores, and the last lane's state is carried into the next load.  A run of kRunLoads loads per wavefront is one "row"
// of level 1; the levels above it reuse reduce_kernel / apply_kernel on the summaries, which
// have the layout of the stream's rows (packed lanes ARE elements).
#ifndef TR_RUN_LOADS
#define TR_RUN_LOADS 16
#endif
#ifndef TR_PREFETCH
#define TR_PREFETCH 1                      // the next load of a run is requested before this one is folded (one more load in flight a wave)
#endif
constexpr int kRunLoads = TR_RUN_LOADS;
constexpr int kRowsPerLane = 4;

// LDS hand-off between lanes of one wavefront (DS ops of a wave execute in issue order)
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// piece p of a load -> LDS slot: rotate within aligned groups of 4 by p/16, so that both "lane l takes
// piece l" and "lane l takes pieces 4l .. 4l+3" spread over the banks
__device__ __forceinline__ uint32_t swz(uint32_t p) { return (p & ~3u) | ((p + (p >> 4)) & 3u); }

struct P16 {                                            // 2 x u16 in a dword: v_pk_add_u16 / v_pk_mul_lo_u16
    typedef uint32_t T;
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ T zero() { return 0; }
    static __device__ __forceinline__ T add(T a, T b) { return __builtin_bit_cast(T, (us2)(__builtin_bit_cast(us2, a) + __builtin_bit_cast(us2, b))); }
    static __device__ __forceinline__ T mul(T a, uint32_t n)
    {
        const us2 nn = {(unsigned short)n, (unsigned short)n};
        return __builtin_bit_cast(T, (us2)(__builtin_bit_cast(us2, a) * nn));
    }
    static __device__ __forceinline__ T shfl(T v, int src) { return (T)__shfl((int)v, src); }
    static constexpr int M = 1;
};
struct P8 {                                             // 4 x u8 in a dword
    typedef uint32_t T;
    static __device__ __forceinline__ T zero() { return 0; }
    static __device__ __forceinline__ T add(T a, T b) { return ((a & 0x7f7f7f7fu) + (b & 0x7f7f7f7fu)) ^ ((a ^ b) & 0x80808080u); }
    static __device__ __forceinline__ T mul(T a, uint32_t n)
    {
        n &= 0xffu;
        return ((a & 0x00ff00ffu) * n & 0x00ff00ffu) | ((((a >> 8) & 0x00ff00ffu) * n & 0x00ff00ffu) << 8);
    }
    static __device__ __forceinline__ T shfl(T v, int src) { return (T)__shfl((int)v, src); }
    static constexpr int M = 1;
};
template <typename U> struct S1x {                      // one element
    typedef U T;
    static __device__ __forceinline__ T zero() { return 0; }
    static __device__ __forceinline__ T add(T a, T b) { return (T)(a + b); }
    static __device__ __forceinline__ T mul(T a, uint32_t n) { return (T)(a * n); }
    static __device__ __forceinline__ T shfl(T v, int src) { return (T)__shfl((int)v, src); }
    static constexpr int M = 1;
};
template <typename P> struct V4 {                       // 4 packed dwords = 16 bytes
    struct __attribute__((aligned(16))) T { uint32_t v[4]; };
    static __device__ __forceinline__ T zero() { return T{{0, 0, 0, 0}}; }
    static __device__ __forceinline__ T add(T a, T b) { T r; for (int k = 0; k < 4; k++) r.v[k] = P::add(a.v[k], b.v[k]); return r; }
    static __device__ __forceinline__ T mul(T a, uint32_t n) { T r; for (int k = 0; k < 4; k++) r.v[k] = P::mul(a.v[k], n); return r; }
    static __device__ __forceinline__ T shfl(T a, int src) { T r; for (int k = 0; k < 4; k++) r.v[k] = P::shfl(a.v[k], src); return r; }
    static constexpr int M = 1;
};
// Rows SHORTER than a 16-byte piece (RB = 1, 2, 4 or 8 bytes: the univariate streams and their few-column
// neighbours): a piece holds M = 16 / RB consecutive rows, so the scan runs over pieces -- a piece's summary is
// the sum over its rows replicated into every row slot, the state is carried replicated -- and the rows inside a
// piece are finished by an in-register prefix (shift by one row and add, log2(M) times).  Same 16-byte requests
// as the wide rows instead of 1-, 2- or 4-byte ones.
template <typename P, int RB> struct Sub : V4<P> {
    typedef typename V4<P>::T T;
    static constexpr int M = 16 / RB;
    static __device__ __forceinline__ uint32_t rep_last(uint32_t a)      // the dword's last row in every row slot of a dword
    {
        if constexpr (RB == 1) return (a >> 24) * 0x01010101u;
        else if constexpr (RB == 2) return (a >> 16) * 0x00010001u;
        else return a;
    }
    static __device__ __forceinline__ T prefix(T a)                      // inclusive prefix over the piece's rows
    {
        if constexpr (RB == 8) { a.v[2] = P::add(a.v[2], a.v[0]); a.v[3] = P::add(a.v[3], a.v[1]); return a; }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if constexpr (RB == 1) { a.v[k] = P::add(a.v[k], a.v[k] << 8); a.v[k] = P::add(a.v[k], a.v[k] << 16); }
            if constexpr (RB == 2) a.v[k] = P::add(a.v[k], a.v[k] << 16);
            if (k > 0) a.v[k] = P::add(a.v[k], rep_last(a.v[k - 1]));
        }
        return a;
    }
    static __device__ __forceinline__ T last(T a)                        // the piece's last row in every row slot
    {
        if constexpr (RB == 8) return T{{a.v[2], a.v[3], a.v[2], a.v[3]}};
        const uint32_t r = rep_last(a.v[3]);
        return T{{r, r, r, r}};
    }
};

// STORE = false: write the run's summary (S1 [, S2]) ; STORE = true: read the run's incoming state and write x
template <typename E, int KIND, bool STORE>
__global__ void __launch_bounds__(kTB) wave_scan_kernel(const typename E::T* y, uint64_t len_e, uint32_t dv, uint64_t nruns,
                                                        const typename E::T* xin, const typename E::T* din, typename E::T* s1_out,
                                                        typename E::T* s2_out, typename E::T* dest, int run_loads)
{
    typedef typename E::T T;
    const uint64_t run = ((uint64_t)blockIdx.x * kTB + threadIdx.x) >> 6;
    if (run >= nruns) return;
    const int lane = threadIdx.x & 63;
    // dv pieces per row, any count <= 64: 64 / dv rows per load, the lanes past the last whole row idle
    const int rpw = 64 / (int)dv;                       // rows per load
    const int r = lane / (int)dv;                       // row of the load
    const int c = lane - r * (int)dv;                   // piece of the row this lane holds
    const bool act = r < rpw;
    const uint32_t PL = (uint32_t)rpw * dv;             // pieces per row-step of a load (64 when dv divides 64)
    T x = E::zero(), d = E::zero();                     // state entering the next load (per column piece)
    if (STORE && xin) {
        x = xin[run * dv + c];
        if (KIND) d = din[run * dv + c];
    }
    constexpr int K = kRowsPerLane;                     // consecutive rows a lane folds in registers before the cross-lane step
    // With 1 or 2 pieces per row a lane's own rows are not next to its neighbours': loading them
    // directly makes every 16-byte piece its own memory request.  So the wavefront loads (and
    // stores) the 64*K pieces in order, lane l piece l, and the rows change hands in LDS.
    __shared__ T xbuf[kTB / 64][64 * K];
    T* const xb = xbuf[threadIdx.x >> 6];
    const bool via_lds = dv < 3;                          // (1 or 2: powers of two, 64 pieces per row-step, which the hand-over is laid out for)
    const uint64_t e0 = run * (uint64_t)run_loads * PL * K;
    // load j's pieces as this lane requests them: piece m * 64 + lane of the load (LDS hand-over) or the lane's own K rows
    auto fetch = [&](int j, T (&v)[K]) {
        const uint64_t eb = e0 + (uint64_t)j * PL * K;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const uint64_t e = via_lds ? eb + (uint64_t)k * 64 + lane : eb + (uint64_t)r * K * dv + (uint64_t)c + (uint64_t)k * dv;
            v[k] = ((via_lds || act) && j < run_loads && e < len_e) ? y[e] : E::zero();
        }
    };
    T nx[K];
    if (TR_PREFETCH) fetch(0, nx);
    for (int j = 0; j < run_loads; j++) {
        const uint64_t eb = e0 + (uint64_t)j * PL * K;  // first piece of this load
        if (eb >= len_e) break;                         // wave-uniform
        const uint64_t el = eb + (uint64_t)r * K * dv + (uint64_t)c;   // this lane's first piece; its rows are dv pieces apart
        T yk[K];
        if (TR_PREFETCH) {
#pragma unroll
            for (int k = 0; k < K; k++) yk[k] = nx[k];
            fetch(j + 1, nx);                           // (past the run's or the stream's end: zeros, no request)
        } else fetch(j, yk);
        T s1 = E::zero(), s2 = E::zero();               // summary of the lane's K rows
        const uint32_t pl = (uint32_t)r * K * dv + (uint32_t)c;      // this lane's first piece within the load
        if (via_lds) {
#pragma unroll
            for (int m = 0; m < K; m++) xb[swz((uint32_t)m * 64 + lane)] = yk[m];
            wave_sync();
#pragma unroll
            for (int k = 0; k < K; k++) yk[k] = xb[swz(pl + (uint32_t)k * dv)];
        }
#pragma unroll
        for (int k = 0; k < K; k++) {
            if constexpr (E::M > 1) {                   // a piece of M rows: (S1, S2, n) o (S1', S2', M)
                const T p1 = E::prefix(yk[k]);
                if (KIND) s2 = E::add(E::add(s2, E::mul(s1, (uint32_t)E::M)), E::last(E::prefix(p1)));
                s1 = E::add(s1, E::last(p1));
            } else {
                s1 = E::add(s1, yk[k]);
                if (KIND) s2 = E::add(s2, s1);
            }
        }
        // inclusive scan of the lane summaries over the rows of the load (stride dv lanes); the right
        // operand of step st covers K * 2^st rows
        T i1 = s1, i2 = s2;
        for (int st = 0; (1 << st) < rpw; st++) {
            const int src = lane - (int)(dv << st);
            const T l1 = E::shfl(i1, src < 0 ? lane : src);
            if (KIND) {
                const T l2 = E::shfl(i2, src < 0 ? lane : src);
                if (r >= (1 << st)) i2 = E::add(E::add(l2, E::mul(l1, (uint32_t)(K * E::M) << st)), i2);
            }
            if (r >= (1 << st)) i1 = E::add(l1, i1);
        }
        // state entering this lane's rows: the carried state advanced over the r*K rows before them
        const int prev = lane - (int)dv;
        T p1 = E::shfl(i1, prev < 0 ? lane : prev), p2 = KIND ? E::shfl(i2, prev < 0 ? lane : prev) : E::zero();
        T xl = x, dl = d;
        if (r > 0) {
            if (KIND) { xl = E::add(E::add(x, E::mul(d, (uint32_t)(r * K * E::M))), p2); dl = E::add(d, p1); }
            else xl = E::add(x, p1);
        }
#pragma unroll
        for (int k = 0; k < K; k++) {
            T outv;
            if constexpr (E::M > 1) {                   // the rows inside the piece: x + (j + 1) d + prefix^2(y), d + prefix(y)
                const T p1 = E::prefix(yk[k]);
                if (KIND) {
                    const T p2 = E::prefix(p1);
                    outv = E::add(E::add(xl, E::prefix(dl)), p2);
                    xl = E::add(E::add(xl, E::mul(dl, (uint32_t)E::M)), E::last(p2));
                    dl = E::add(dl, E::last(p1));
                } else {
                    outv = E::add(xl, p1);
                    xl = E::add(xl, E::last(p1));
                }
            } else {
                if (KIND) { dl = E::add(dl, yk[k]); xl = E::add(xl, dl); }
                else xl = E::add(xl, yk[k]);
                outv = xl;
            }
            const uint64_t e = el + (uint64_t)k * dv;
            if (STORE) {
                if (via_lds) xb[swz(pl + (uint32_t)k * dv)] = outv;
                else if (act && e < len_e) dest[e] = outv;
            }
        }
        if (STORE && via_lds) {
            wave_sync();
#pragma unroll
            for (int m = 0; m < K; m++) {
                const uint64_t e = eb + (uint64_t)m * 64 + lane;
                if (e < len_e) dest[e] = xb[swz((uint32_t)m * 64 + lane)];
            }
        }
        if (via_lds) wave_sync();
        const int last = (rpw - 1) * (int)dv + c;       // the lane holding the load's last rows of this column piece
        x = E::shfl(xl, last);
        if (KIND) d = E::shfl(dl, last);
    }
    if (!STORE && r == 0) {                             // the run's summary: the state it turns (0, 0) into
        s1_out[run * dv + c] = KIND ? d : x;
        if (KIND) s2_out[run * dv + c] = x;
    }
}


// ---------------------------------------------------------------- decode in ONE pass over the stream (round 6; VERDICT r5 "next" 8)
// The two wave_scan launches read the stream twice (summaries, then the store pass): 3 bytes moved for every 2 of the job, and both passes
// run at the memory system's rate (prefetching the next load, runs of 4 / 8 / 16 loads: nothing, tools/tr_ab.sh).  Here a workgroup keeps its
// TILE (kChainWaves waves x kChainLoads loads of 64 x kRowsPerLane pieces: 128 KB at 16-byte pieces) in registers between the two: it folds the
// tile, publishes the tile's summary, gets the state ENTERING the tile from a chained scan over the tiles before it (decoupled look-back:
// tiles are taken in ticket order, a tile waits only for lower tickets -- which are running), publishes the state leaving it, and finishes
// and stores its rows.
// What makes the look-back cheap here where four forms of it lost in the encoders: the tiles are large (a few hundred in flight
// on the chip) and the workgroup looks back with ALL its waves at once -- wave w takes the 64 / dv tiles at distance w * 64 / dv + .. (a
// lane per tile and column piece), composes them from its nearest published STATE on with a shuffle tree, the waves' partial results meet in
// LDS -- so that one step covers 512 / dv predecessors: every tile older than the ones in flight has its state out, one step finds it.
// The composition is the one at the top of this file; a published state (x, d) is the summary (S1, S2) = (d, x) of everything before it.
// Column pieces are independent scans: each finds its own nearest state.
#ifndef TR_CHAIN_LOADS
#define TR_CHAIN_LOADS 4
#endif
#ifndef TR_CHAIN_PREFETCH
#define TR_CHAIN_PREFETCH 1
#endif
#ifndef TR_CHAIN_WAVES
#define TR_CHAIN_WAVES 8
#endif
constexpr int kChainWaves = TR_CHAIN_WAVES, kChainT = 64 * kChainWaves, kChainLoads = TR_CHAIN_LOADS, kChainMaxDv = 8;

// What crosses workgroups: 64-bit words of 32 data bits under a 32-bit tag (0: not there yet -- the scratch is zeroed before the launch),
// written and read with relaxed agent-scope atomics, summary and state in arrays of their own.  A word is there or not, whole: no flag beside
// the data, no fence, no wait between a tile's data and its flag.  (The first form had a flag per tile behind an agent-scope release fence,
// which writes the XCD's whole L2 back -- the tile's own output sits dirty in it -- and readers behind an acquire, which invalidates it:
// 0.167 ms where the two-pass decode takes 0.092.  With the data in atomics and an s_waitcnt in front of the flag: 0.094, of which a tile
// spent 1.6 us of its 19.7 publishing and 4.0 looking back through two dependent round trips, tools/chain_phases.py.)
template <typename T> struct ChainWs {
    static constexpr int kWords = sizeof(T) == 16 ? 4 : 1;
    uint32_t* ticket;
    uint64_t* agg1; uint64_t* agg2;          // a tile's summary, dv pieces of kWords words each
    uint64_t* inc1; uint64_t* inc2;          // the state leaving a tile
};
template <typename T> __device__ __forceinline__ void chain_put(uint64_t* base, uint64_t i, const T& v)
{
    constexpr int NW = ChainWs<T>::kWords;
    uint32_t d[NW];
    if constexpr (sizeof(T) == 16) __builtin_memcpy(d, &v, 16);
    else d[0] = (uint32_t)v;
#pragma unroll
    for (int k = 0; k < NW; k++) __hip_atomic_store(base + i * NW + k, (1ull << 32) | d[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T> __device__ __forceinline__ bool chain_try(uint64_t* base, uint64_t i, T& v)
{
    constexpr int NW = ChainWs<T>::kWords;
    uint64_t w[NW];
#pragma unroll
    for (int k = 0; k < NW; k++) w[k] = __hip_atomic_load(base + i * NW + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t d[NW];
    bool ok = true;
#pragma unroll
    for (int k = 0; k < NW; k++) { ok = ok && (w[k] >> 32) != 0ull; d[k] = (uint32_t)w[k]; }
    if constexpr (sizeof(T) == 16) __builtin_memcpy(&v, d, 16);
    else v = (T)d[0];
    return ok;
}

// one piece a row (dv == 1): the scans across the lanes as DPP moves (row_shr 1 / 2 / 4 / 8, row_bcast 15 / 31: lanes without a source read 0, the
// identity) instead of ds_bpermute round trips
template <int CTRL, int RM> __device__ __forceinline__ uint32_t dpp0(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, RM, 0xf, false); }
template <typename T, int CTRL, int RM> __device__ __forceinline__ T dpp_t(const T& v)
{
    if constexpr (sizeof(T) == 16) {
        T r;
#pragma unroll
        for (int k = 0; k < 4; k++) r.v[k] = dpp0<CTRL, RM>(v.v[k]);
        return r;
    } else {
        return (T)dpp0<CTRL, RM>((uint32_t)v);
    }
}
template <typename T> __device__ __forceinline__ T lane63_t(const T& v)
{
    if constexpr (sizeof(T) == 16) {
        T r;
#pragma unroll
        for (int k = 0; k < 4; k++) r.v[k] = (uint32_t)__builtin_amdgcn_readlane((int)v.v[k], 63);
        return r;
    } else {
        return (T)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63);
    }
}

#ifdef TR_CHAIN_TIMING                         // experiment builds: where a tile's time goes (thread 0 of every tile < 4096; 10 ns ticks)
__device__ uint64_t g_chain_ts[4096][8];
#define CHAIN_TS(k) do { if (threadIdx.x == 0 && tile < 4096u) g_chain_ts[tile][k] = wall_clock64(); } while (0)
#else
#define CHAIN_TS(k) do {} while (0)
#endif
#ifndef TR_CHAIN_EU0
#define TR_CHAIN_EU0 2                      // waves a SIMD the compiler is asked to leave room for: delta / double delta
#endif
#ifndef TR_CHAIN_EU1
#define TR_CHAIN_EU1 2
#endif
template <typename E, int KIND>
__global__ void __launch_bounds__(kChainT) __attribute__((amdgpu_waves_per_eu(KIND ? TR_CHAIN_EU1 : TR_CHAIN_EU0))) chain_scan_kernel(const typename E::T* y, uint64_t len_e, uint32_t dv, ChainWs<typename E::T> ws,
                                                             typename E::T* dest, uint32_t ntiles)
{
    typedef typename E::T T;
    constexpr int K = kRowsPerLane, J = kChainLoads;
    __shared__ T xbuf[kChainWaves][64 * K];
    __shared__ T sm1[kChainWaves][kChainMaxDv], sm2[kChainWaves][kChainMaxDv];
    __shared__ uint32_t smn[kChainWaves][kChainMaxDv], sminc[kChainWaves][kChainMaxDv], s_ticket;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) s_ticket = __hip_atomic_fetch_add(ws.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    uint32_t tile = s_ticket;
    __syncthreads();
    const int rpw = 64 / (int)dv, r = lane / (int)dv, c = lane - r * (int)dv;
    const bool act = r < rpw;
    const uint32_t PL = (uint32_t)rpw * dv;
    const bool via_lds = dv < 3;
    T* const xb = xbuf[w];
    const uint32_t nL = (uint32_t)(rpw * K * E::M), nW = (uint32_t)J * nL, nT = (uint32_t)kChainWaves * nW;      // rows a load / a wave / a tile
    const uint32_t nR = (uint32_t)(r * K * E::M);                            // rows of a load in front of this lane's
    const uint32_t pl = (uint32_t)r * K * dv + (uint32_t)c;                  // this lane's first piece within a load
    const int last = (rpw - 1) * (int)dv + c;                                // the lane holding a load's last rows of this column piece
    const int prev = lane - (int)dv < 0 ? lane : lane - (int)dv;

    // A workgroup takes tiles until there are none (one workgroup a CU: the tile's registers).  Delta: it asks for its NEXT tile's loads before it
    // looks back for this one -- the memory system, idle while a lone workgroup folds, publishes and looks back (7 of a tile's 13.6 us,
    // tools/chain_phases.py), has the next tile to fetch meanwhile: 0.0716 -> 0.0654 ms at 64 Mi samples.  It is not the whole of the idle time that
    // comes back: this hardware counts a wave's loads in order, so the look-back's polls wait for every load requested before them (look-back
    // 4.5 -> 6.9 us, the wait for the tile 3.8 -> 1.2); requested at the tile's top instead, two tickets ahead, the top waits for the tile before's
    // stores: 0.0729 ms (0.470 against 0.491 at 512 Mi).  Double delta has no registers left for a second tile (spills: 0.089 -> 0.118 ms): no prefetch.
    constexpr bool PF = KIND == 0 && TR_CHAIN_PREFETCH;
    T nk[J][K];
    auto request = [&](uint32_t t) {                                       // every load of the wave's part of tile t, as the lanes request them
        const uint64_t f0 = ((uint64_t)t * kChainWaves + (uint64_t)w) * (uint64_t)J * PL * K;
#pragma unroll
        for (int j = 0; j < J; j++) {
            const uint64_t eb = f0 + (uint64_t)j * PL * K;
#pragma unroll
            for (int k = 0; k < K; k++) {
                const uint64_t e = via_lds ? eb + (uint64_t)k * 64 + lane : eb + (uint64_t)pl + (uint64_t)k * dv;
                nk[j][k] = ((via_lds || act) && e < len_e) ? y[e] : E::zero();
            }
        }
    };
    if (PF) request(tile);
    while (tile < ntiles) {
    CHAIN_TS(0);
    if (tid == 0) s_ticket = __hip_atomic_fetch_add(ws.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // the next tile (read behind the first barrier below)
    if (!PF) request(tile);
    const uint64_t e0 = ((uint64_t)tile * kChainWaves + (uint64_t)w) * (uint64_t)J * PL * K;

    // ---- the tile: (1 or 2 pieces a row) handed over in LDS to the lanes that fold them
    T yk[J][K];
#pragma unroll
    for (int j = 0; j < J; j++)
#pragma unroll
        for (int k = 0; k < K; k++) yk[j][k] = nk[j][k];
    if (via_lds) {
#pragma unroll
        for (int j = 0; j < J; j++) {
#pragma unroll
            for (int m = 0; m < K; m++) xb[swz((uint32_t)m * 64 + lane)] = yk[j][m];
            wave_sync();
#pragma unroll
            for (int k = 0; k < K; k++) yk[j][k] = xb[swz(pl + (uint32_t)k * dv)];
            wave_sync();
        }
    }
    CHAIN_TS(1);
    // (S1, S2, n) o (S1', S2', n'): `o` the older operand, (a, n) the newer one, result in a
    auto compose = [&](const T& o1, const T& o2, uint32_t on, T& a1, T& a2, uint32_t& an) {
        if (KIND) a2 = E::add(E::add(o2, E::mul(o1, an)), a2);
        a1 = E::add(o1, a1);
        an += on;
    };

    // ---- the lanes' summaries of every load, the J scans across the lanes side by side (a scan step is a dependent LDS round trip: one after
    // the other they were 3.3 of a tile's 19.7 us), then (q1, q2)[j] = the summary of the wave's rows in front of this lane's rows of load j
    T q1[J], q2[J], W1, W2;
    {
        T i1[J], i2[J];
#pragma unroll
        for (int j = 0; j < J; j++) {
            T s1 = E::zero(), s2 = E::zero();
#pragma unroll
            for (int k = 0; k < K; k++) {
                if constexpr (E::M > 1) {                   // a piece of M rows: (S1, S2, n) o (S1', S2', M)
                    const T p1 = E::prefix(yk[j][k]);
                    if (KIND) s2 = E::add(E::add(s2, E::mul(s1, (uint32_t)E::M)), E::last(E::prefix(p1)));
                    s1 = E::add(s1, E::last(p1));
                } else {
                    s1 = E::add(s1, yk[j][k]);
                    if (KIND) s2 = E::add(s2, s1);
                }
            }
            i1[j] = s1;
            i2[j] = s2;
        }
        const bool dpp_scan = dv == 1u;
        if (dpp_scan) {
            // (the left operand's S1 is multiplied by the rows of the RIGHT one: 2^st lanes' where a lane has a source; in the two broadcast
            //  steps what the lane has gathered so far -- its place in the row of 16, of 32)
#define CHAIN_DPP_STEP(CTRL, RM, NR)                                                                                          \
            _Pragma("unroll") for (int j = 0; j < J; j++) {                                                                   \
                const T l1 = dpp_t<T, CTRL, RM>(i1[j]);                                                                       \
                if (KIND) { const T l2 = dpp_t<T, CTRL, RM>(i2[j]); i2[j] = E::add(E::add(l2, E::mul(l1, (uint32_t)(NR))), i2[j]); } \
                i1[j] = E::add(l1, i1[j]);                                                                                    \
            }
            CHAIN_DPP_STEP(0x111, 0xf, K * E::M)
            CHAIN_DPP_STEP(0x112, 0xf, 2 * K * E::M)
            CHAIN_DPP_STEP(0x114, 0xf, 4 * K * E::M)
            CHAIN_DPP_STEP(0x118, 0xf, 8 * K * E::M)
            CHAIN_DPP_STEP(0x142, 0xa, ((lane & 15) + 1) * K * E::M)
            CHAIN_DPP_STEP(0x143, 0xc, ((lane & 31) + 1) * K * E::M)
#undef CHAIN_DPP_STEP
        } else
        for (int st = 0; (1 << st) < rpw; st++) {
            const int src = lane - (int)(dv << st);
            const bool on = r >= (1 << st);
#pragma unroll
            for (int j = 0; j < J; j++) {
                const T l1 = E::shfl(i1[j], src < 0 ? lane : src);
                if (KIND) {
                    const T l2 = E::shfl(i2[j], src < 0 ? lane : src);
                    if (on) i2[j] = E::add(E::add(l2, E::mul(l1, (uint32_t)(K * E::M) << st)), i2[j]);
                }
                if (on) i1[j] = E::add(l1, i1[j]);
            }
        }
        T e1 = E::zero(), e2 = E::zero();                   // the loads before j
#pragma unroll
        for (int j = 0; j < J; j++) {
            T p1, p2 = E::zero(), t1, t2 = E::zero();
            if (dpp_scan) {                                 // wave_shr:1; lane 63
                p1 = dpp_t<T, 0x138, 0xf>(i1[j]);
                t1 = lane63_t<T>(i1[j]);
                if (KIND) { p2 = dpp_t<T, 0x138, 0xf>(i2[j]); t2 = lane63_t<T>(i2[j]); }
            } else {
                p1 = E::shfl(i1[j], prev);
                t1 = E::shfl(i1[j], last);
                if (KIND) { p2 = E::shfl(i2[j], prev); t2 = E::shfl(i2[j], last); }
            }
            q1[j] = e1;
            q2[j] = e2;
            if (r > 0) {
                q1[j] = E::add(e1, p1);
                if (KIND) q2[j] = E::add(E::add(e2, E::mul(e1, nR)), p2);
            }
            if (KIND) e2 = E::add(E::add(e2, E::mul(e1, nL)), t2);
            e1 = E::add(e1, t1);
        }
        W1 = e1;
        W2 = e2;
    }
    CHAIN_TS(2);
    if (lane < (int)dv) { sm1[w][lane] = W1; if (KIND) sm2[w][lane] = W2; }
    __syncthreads();
    const uint32_t next = s_ticket;
    T B1 = E::zero(), B2 = E::zero(), T1 = E::zero(), T2 = E::zero();     // the waves before mine; the whole tile
    {
        uint32_t bn = 0, tn = 0;
#pragma unroll
        for (int k = kChainWaves - 1; k >= 0; k--) {                      // newest first: compose(older, acc)
            const T a1 = sm1[k][c], a2 = KIND ? sm2[k][c] : E::zero();
            compose(a1, a2, nW, T1, T2, tn);
            if (k < w) compose(a1, a2, nW, B1, B2, bn);
        }
    }
    __syncthreads();                                                      // (sm1 / sm2 are the look-back's from here on)
    const uint64_t mine = (uint64_t)tile * dv + (uint64_t)c;
    if (w == 0 && lane < (int)dv) {
        chain_put<T>(tile == 0 ? ws.inc1 : ws.agg1, mine, T1);
        if (KIND) chain_put<T>(tile == 0 ? ws.inc2 : ws.agg2, mine, T2);
    }
    CHAIN_TS(3);
    if (PF && next < ntiles) request(next);

    // ---- the state entering the tile: (A1, A2) = everything before it
    T A1 = E::zero(), A2 = E::zero();
    uint32_t An = 0;
    if (tile != 0) {
        uint64_t colmask = 0;                                             // the lanes of my column piece
        for (int rr = 0; rr < rpw; rr++) colmask |= 1ull << (rr * (int)dv + c);
        bool done = false;                                                // my column piece has found its nearest state
        int64_t base = (int64_t)tile - 1;
        for (;;) {
            const int64_t p = base - ((int64_t)w * rpw + r);               // this lane's tile
            uint32_t f = 2u;                                               // before tile 0: the zero state; a column that is done: nothing
            T v1 = E::zero(), v2 = E::zero();
            if (act && p >= 0 && !done) {
                const uint64_t at = (uint64_t)p * dv + (uint64_t)c;
                for (;;) {                                                 // its holder is running (tickets): this ends
                    // (state and summary asked for together: one round trip whichever is there)
                    T a = E::zero(), b = E::zero(), g = E::zero(), h = E::zero();
                    bool oki = chain_try<T>(ws.inc1, at, a), oka = chain_try<T>(ws.agg1, at, g);
                    if (KIND) { oki = chain_try<T>(ws.inc2, at, b) && oki; oka = chain_try<T>(ws.agg2, at, h) && oka; }
                    if (oki) { f = 2u; v1 = a; v2 = b; break; }
                    if (oka) { f = 1u; v1 = g; v2 = h; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            const uint64_t inc_lanes = __ballot(act && f == 2u) & colmask;
            const int first_r = inc_lanes ? (int)__builtin_ctzll(inc_lanes) / (int)dv : rpw;      // the nearest tile of my window that has its state out
            const bool use = act && r <= first_r && p >= 0 && !done;
            if (!use) { v1 = E::zero(); v2 = E::zero(); }
            uint32_t vn = use ? nT : 0u;
            for (int st = 0; (1 << st) < rpw; st++) {                      // lane r <- (lane r + 2^st: older) o (lane r)
                const int src = lane + (int)(dv << st);
                const bool ok = r + (1 << st) < rpw;
                const T o1 = E::shfl(v1, ok ? src : lane), o2 = KIND ? E::shfl(v2, ok ? src : lane) : E::zero();
                const uint32_t on = (uint32_t)__shfl((int)vn, ok ? src : lane);
                if (ok) compose(o1, o2, on, v1, v2, vn);
            }
            if (lane < (int)dv) { sm1[w][lane] = v1; if (KIND) sm2[w][lane] = v2; smn[w][lane] = vn; sminc[w][lane] = inc_lanes != 0ull ? 1u : 0u; }
            __syncthreads();
            T P1 = E::zero(), P2 = E::zero();
            uint32_t Pn = 0;
            bool found = false;
            for (int k = 0; k < kChainWaves && !found; k++) {              // wave 0's window is the nearest
                compose(sm1[k][c], KIND ? sm2[k][c] : E::zero(), smn[k][c], P1, P2, Pn);
                found = sminc[k][c] != 0u;
            }
            if (!done) compose(P1, P2, Pn, A1, A2, An);
            done = done || found;
            if (__syncthreads_and(done ? 1 : 0)) break;
            base -= (int64_t)kChainWaves * rpw;
        }
        // the state leaving the tile
        if (w == 0 && lane < (int)dv) {
            T I1 = T1, I2 = T2;
            uint32_t In = nT;
            compose(A1, A2, An, I1, I2, In);
            chain_put<T>(ws.inc1, mine, I1);
            if (KIND) chain_put<T>(ws.inc2, mine, I2);
        }
    }
    CHAIN_TS(4);

    // ---- my wave's rows: the state entering the tile, advanced over the waves before mine, then over the wave's rows in front of each lane's
    {
        uint32_t bn = (uint32_t)w * nW;
        compose(A1, A2, An, B1, B2, bn);
        const T x = KIND ? B2 : B1, d = KIND ? B1 : E::zero();
#pragma unroll
        for (int j = 0; j < J; j++) {
            const uint64_t eb = e0 + (uint64_t)j * PL * K;
            T xl, dl = E::zero();
            if (KIND) { xl = E::add(E::add(x, E::mul(d, (uint32_t)j * nL + nR)), q2[j]); dl = E::add(d, q1[j]); }
            else xl = E::add(x, q1[j]);
#pragma unroll
            for (int k = 0; k < K; k++) {
                T outv;
                if constexpr (E::M > 1) {                   // the rows inside the piece: x + (j + 1) d + prefix^2(y), d + prefix(y)
                    const T g1 = E::prefix(yk[j][k]);
                    if (KIND) {
                        const T g2 = E::prefix(g1);
                        outv = E::add(E::add(xl, E::prefix(dl)), g2);
                        xl = E::add(E::add(xl, E::mul(dl, (uint32_t)E::M)), E::last(g2));
                        dl = E::add(dl, E::last(g1));
                    } else {
                        outv = E::add(xl, g1);
                        xl = E::add(xl, E::last(g1));
                    }
                } else {
                    if (KIND) { dl = E::add(dl, yk[j][k]); xl = E::add(xl, dl); }
                    else xl = E::add(xl, yk[j][k]);
                    outv = xl;
                }
                const uint64_t e = eb + (uint64_t)pl + (uint64_t)k * dv;
                if (via_lds) xb[swz(pl + (uint32_t)k * dv)] = outv;
                else if (act && e < len_e) dest[e] = outv;
            }
            if (via_lds) {
                wave_sync();
#pragma unroll
                for (int m = 0; m < K; m++) {
                    const uint64_t e = eb + (uint64_t)m * 64 + lane;
                    if (e < len_e) dest[e] = xb[swz((uint32_t)m * 64 + lane)];
                }
                wave_sync();
            }
        }
    }
    CHAIN_TS(5);
    tile = next;
    }
}

}  // namespace
namespace sprintz { int set_error(int code, const char* what); }   // api.hip: the library's one error sink
namespace {
int fail(int code, const char* what) { return sprintz::set_error(code, what); }

#ifndef TR_CHAIN_WGS_PER_CU
#define TR_CHAIN_WGS_PER_CU 1
#endif
#ifndef TR_CHAIN_MIN_TILES
#define TR_CHAIN_MIN_TILES 1                  // (from the first tile on: 9.6 against 13.7 us at 32 KB, 13 against 52 at 0.5 MB, 64 against 90 at 128 MB -- tools/chain_sizes.py)
#endif
// scratch of the one-pass decode: the ticket + two (delta) or four arrays of dv x 1 or 4 tagged 64-bit words per tile.  A tile is >= 8 waves
// x 2 loads x 56 pieces x 4 rows, dv <= 8: below 12 % of the stream for 1-byte pieces (unaligned streams), 1.5 % for 16-byte ones
size_t chain_tmp_bound(uint64_t stream_bytes) { return (size_t)(stream_bytes / 8) + 8192; }
// workgroups of the one-pass decode: one per CU (each takes tiles until there are none)
int chain_resident_wgs()
{
    static const int n = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        return cus * TR_CHAIN_WGS_PER_CU;
    }();
    return n;
}
// SPRINTZ_MI355X_TRANSFORM_CHAIN: 0 = the two-pass decode always; n > 0 = the one-pass decode from n tiles on (default TR_CHAIN_MIN_TILES;
// 1 lets the tests drive the chained scan with small streams)
uint64_t chain_min_tiles()
{
    const char* e = getenv("SPRINTZ_MI355X_TRANSFORM_CHAIN");
    if (!e || !e[0]) return TR_CHAIN_MIN_TILES;
    const long v = strtol(e, nullptr, 10);
    return v <= 0 ? ~0ull : (uint64_t)v;
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

// levels above `base` (base.rows runs of base.span original rows each, summaries already in base.s1/s2):
// fills base.xin / base.din with every run's incoming state.  With base = the input itself
// (span 1) the last apply writes x into dest instead.
template <typename U, int KIND>
int scan_levels(Level<U> base, uint64_t len, uint32_t D, uint64_t rows0, U* dest, uint8_t* tmp, hipStream_t st)
{
    std::vector<Level<U>> lv;
    lv.push_back(base);
    uint8_t* t = tmp;
    auto take = [&](uint64_t n) { U* q = (U*)t; t += (n * D * sizeof(U) + 63) & ~(size_t)63; return q; };
    while (lv.back().rows > 1) {
        const uint64_t rows = (lv.back().rows + R - 1) / R;
        U* s1 = take(rows);
        U* s2 = KIND ? take(rows) : nullptr;
        U* xi = take(rows);
        U* di = KIND ? take(rows) : nullptr;
        lv.push_back(Level<U>{s1, s2, xi, di, rows, lv.back().span * R});
    }
    const size_t L = lv.size();
    for (size_t k = 0; k + 1 < L; k++) {
        const uint64_t threads = lv[k + 1].rows * D;
        hipLaunchKernelGGL((reduce_kernel<U, KIND>), dim3((unsigned)((threads + kTB - 1) / kTB)), dim3(kTB), 0, st, lv[k], len, D, rows0,
                           (U*)lv[k + 1].s1, (U*)lv[k + 1].s2, lv[k + 1].rows);
    }
    if (L == 1) {                                        // a single run: one pseudo-run above it, entering with the zero state
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

// piece size of the wave-scan path for this stream: 16, 4 or sizeof(U) bytes -- the largest that
// divides the row, the stream and the pointers' alignment, with a power-of-two piece count <= 64
template <typename U>
int wave_piece_bytes(const void* y, const void* dest, uint64_t len, uint32_t D)
{
    const uint64_t row_bytes = (uint64_t)D * sizeof(U), total = len * sizeof(U);
    for (int pb : {16, 4, (int)sizeof(U)}) {
        if (pb < (int)sizeof(U) || row_bytes % pb || total % pb) continue;
        if (((uintptr_t)y | (uintptr_t)dest) % pb) continue;
        const uint64_t dv = row_bytes / pb;
        // any piece count up to 64 (64 / dv whole rows per load); 1 and 2 go through the LDS hand-over, which is laid out for 64 pieces
        if (dv <= 64 && ((dv & (dv - 1)) == 0 || dv >= 3)) return pb;
    }
    return 0;
}

template <typename U, typename E, int KIND>
int decode_wave(const U* y, uint64_t len, uint32_t D_real, U* dest, uint8_t* tmp, hipStream_t st)
{
    typedef typename E::T T;
    // (rows shorter than a piece, E::M > 1: the levels above see the replicated 16-byte pieces as rows of 16 / sizeof(U) columns)
    const uint32_t D = E::M > 1 ? (uint32_t)(sizeof(T) / sizeof(U)) : D_real;
    const uint32_t dv = E::M > 1 ? 1u : (uint32_t)((uint64_t)D * sizeof(U) / sizeof(T));
    const uint64_t len_e = len * sizeof(U) / sizeof(T);
    const uint64_t rows0 = (len + D_real - 1) / D_real;                  // real rows
    // loads a run: 16 where there are runs enough to fill the chip; short streams take 4 or 1 -- a run's loads are a serial chain, and a stream of
    // eight 16-load runs was two passes of 16 dependent round trips on eight waves: 52 us for 0.5 MB (tools/chain_sizes.py) -- as long as the
    // runs' states still come from the one workgroup of scan_runs_kernel (<= 4 096 runs)
    int run_loads = kRunLoads;
    {
        const uint64_t rows16 = (uint64_t)kRunLoads * kRowsPerLane * (64 / dv) * E::M, n16 = (rows0 + rows16 - 1) / rows16;
        if (n16 * 16 <= 4096) run_loads = 1;
        else if (n16 * 4 <= 4096) run_loads = 4;
    }
    const uint64_t run_rows = (uint64_t)run_loads * kRowsPerLane * (64 / dv) * E::M;
    const uint64_t nruns = (rows0 + run_rows - 1) / run_rows;
    const unsigned grid = (unsigned)((nruns * 64 + kTB - 1) / kTB);
    // ---- streams of at least TR_CHAIN_MIN_TILES tiles, rows of at most kChainMaxDv pieces: one pass (chain_scan_kernel)
    // (16-bit elements only: at 8 bits the packed arithmetic is five instructions an add where v_pk_add_u16 is one, the tile's fold and walk
    //  run on one workgroup a CU, and the form measured SLOWER than the two passes on every column count -- tools/transforms_shapes.py:
    //  delta 0.126 - 0.167 against 0.103 - 0.143 ms on 128 MiB, double delta 0.28 - 0.48 against 0.15 - 0.26; at 16 bits 0.063 - 0.074 against 0.090 - 0.124.
    //  And 16-byte pieces only: with 4- or 2-byte pieces (rows that are not multiples of 16 bytes) a tile is 32 / 16 KB and the look-back is the kernel:
    //  3 / 5 / 7 / 10 / 12 uint16 columns 0.199 / 0.284 / 0.220 / 0.147 / 0.133 against 0.160 / 0.158 / 0.154 / 0.125 / 0.121 ms.)
    if constexpr (sizeof(U) == 2 && sizeof(T) == 16)
    if (dv <= (uint32_t)kChainMaxDv) {
        const uint64_t tile_e = (uint64_t)kChainWaves * kChainLoads * (uint64_t)((64 / dv) * dv) * kRowsPerLane;
        const uint64_t ntiles = (len_e + tile_e - 1) / tile_e;
        auto up = [](size_t v, size_t a) { return (v + a - 1) & ~(a - 1); };
        const size_t arr = up((size_t)ntiles * dv * ChainWs<T>::kWords * 8, 256), need = 256 + (KIND ? 4 : 2) * arr;
        if (ntiles >= chain_min_tiles() && ntiles < 0x7fffffffull && need <= chain_tmp_bound(len * sizeof(U))) {
            ChainWs<T> ws;
            ws.ticket = (uint32_t*)tmp;
            ws.agg1 = (uint64_t*)(tmp + 256);
            ws.inc1 = (uint64_t*)(tmp + 256 + arr);
            ws.agg2 = KIND ? (uint64_t*)(tmp + 256 + 2 * arr) : nullptr;
            ws.inc2 = KIND ? (uint64_t*)(tmp + 256 + 3 * arr) : nullptr;
            if (hipMemsetAsync(tmp, 0, need, st) != hipSuccess) return fail(SPRINTZ_E_HIP, "transform decode: hipMemsetAsync of the tiles' words");
            const uint64_t wgs = ntiles < (uint64_t)chain_resident_wgs() ? ntiles : (uint64_t)chain_resident_wgs();
            hipLaunchKernelGGL((chain_scan_kernel<E, KIND>), dim3((unsigned)wgs), dim3(kChainT), 0, st, (const T*)y, len_e, dv, ws, (T*)dest, (uint32_t)ntiles);
            return hipGetLastError() == hipSuccess ? 0 : fail(SPRINTZ_E_HIP, "transform decode launch");
        }
    }
    if (nruns == 1) {
        hipLaunchKernelGGL((wave_scan_kernel<E, KIND, true>), dim3(grid), dim3(kTB), 0, st, (const T*)y, len_e, dv, nruns,
                           (const T*)nullptr, (const T*)nullptr, (T*)nullptr, (T*)nullptr, (T*)dest, run_loads);
        return hipGetLastError() == hipSuccess ? 0 : fail(SPRINTZ_E_HIP, "transform decode launch");
    }
    // level 1 = one summary per run, laid out like rows of the stream
    uint8_t* t = tmp;
    auto take = [&](uint64_t n) { U* q = (U*)t; t += (n * D * sizeof(U) + 63) & ~(size_t)63; return q; };
    U* s1 = take(nruns);
    U* s2 = KIND ? take(nruns) : nullptr;
    U* xi = take(nruns);
    U* di = KIND ? take(nruns) : nullptr;
    hipLaunchKernelGGL((wave_scan_kernel<E, KIND, false>), dim3(grid), dim3(kTB), 0, st, (const T*)y, len_e, dv, nruns, (const T*)nullptr,
                       (const T*)nullptr, (T*)s1, (T*)s2, (T*)nullptr, run_loads);
    const Level<U> base{s1, s2, xi, di, nruns, run_rows};
    if (D <= 64 && nruns <= 4096) {                      // (wide rows, or streams long enough that six ~4.6 us launches do not matter: the generic levels --
                                                         //  with 16 384 runs the one workgroup measured slower than they are: 0.72 against 0.66 ms at 512 Mi samples)
        const uint32_t Gmax = (uint32_t)kTopT / D, G = (uint32_t)(nruns < Gmax ? nruns : Gmax);
        hipLaunchKernelGGL((scan_runs_kernel<U, KIND>), dim3(1), dim3(kTopT), 0, st, base, D, rows0, G);
    } else {
        int rc = scan_levels<U, KIND>(base, len, D, rows0, dest, t, st);
        if (rc) return rc;
    }
    hipLaunchKernelGGL((wave_scan_kernel<E, KIND, true>), dim3(grid), dim3(kTB), 0, st, (const T*)y, len_e, dv, nruns, (const T*)xi,
                       (const T*)di, (T*)nullptr, (T*)nullptr, (T*)dest, run_loads);
    return hipGetLastError() == hipSuccess ? 0 : fail(SPRINTZ_E_HIP, "transform decode launch");
}

template <typename U, int KIND>
int decode_device(const U* y, uint64_t len, uint32_t D, U* dest, uint8_t* tmp, hipStream_t st)
{
    typedef typename std::conditional<sizeof(U) == 1, P8, P16>::type P;
    const uint64_t rb = (uint64_t)D * sizeof(U);
    if (rb < 16 && 16 % rb == 0 && (len * sizeof(U)) % 16 == 0 && (((uintptr_t)y | (uintptr_t)dest) % 16) == 0) {
        switch ((int)rb) {                               // rows shorter than a piece
            case 1: if constexpr (sizeof(U) == 1) return decode_wave<U, Sub<P, 1>, KIND>(y, len, D, dest, tmp, st); else break;
            case 2: return decode_wave<U, Sub<P, 2>, KIND>(y, len, D, dest, tmp, st);
            case 4: return decode_wave<U, Sub<P, 4>, KIND>(y, len, D, dest, tmp, st);
            case 8: return decode_wave<U, Sub<P, 8>, KIND>(y, len, D, dest, tmp, st);
            default: break;
        }
    }
    switch (wave_piece_bytes<U>(y, dest, len, D)) {
        case 16: return decode_wave<U, V4<P>, KIND>(y, len, D, dest, tmp, st);
        case 4: return decode_wave<U, P, KIND>(y, len, D, dest, tmp, st);
        case 1: case 2: return decode_wave<U, S1x<U>, KIND>(y, len, D, dest, tmp, st);
        default: break;
    }
    const uint64_t rows0 = (len + D - 1) / D;
    return scan_levels<U, KIND>(Level<U>{y, y, nullptr, nullptr, rows0, 1}, len, D, rows0, dest, tmp, st);
}

template <typename U, int KIND>
int encode_device(const U* x, uint64_t len, uint32_t D, U* y, hipStream_t st)
{
    constexpr int V = 16 / sizeof(U);
    const uint64_t threads = (len + V - 1) / V;
    hipLaunchKernelGGL((encode_kernel<U, KIND>), dim3((unsigned)((threads + kTB - 1) / kTB)), dim3(kTB), 0, st, x, len, D, y);
    return hipGetLastError() == hipSuccess ? 0 : fail(SPRINTZ_E_HIP, "transform encode launch");
}

// ---------------------------------------------------------------- xff: a lane per column (predict.cpp:121-262, :372-497)
template <int W> __device__ __forceinline__ int sx(int v) { return W == 8 ? (int)(int8_t)v : (int)(int16_t)v; }

template <typename U, bool DECODE>
__global__ void __launch_bounds__(64) xff_kernel(const U* __restrict__ in, uint64_t len, uint32_t D, uint32_t nblocks, U* __restrict__ out)
{
    constexpr int W = 8 * (int)sizeof(U);
    const uint32_t col = blockIdx.x * 64u + threadIdx.x;
    if (col >= D) return;
    const bool odd_col = (col & 1u) != 0;
    uint32_t pv = 0;
    int pd = 0, ctr = 0;                                  // ctr: int16 at 8 bits, int32 at 16 (predict.cpp:84-87)
    const U* p = in + col;
    U* q = out + col;
    // the next block's samples are in flight while one is forecast.  (Fetching 8 blocks ahead buys
    // nothing: the walk is bound by ONE wave's dependent-instruction latency, ~20 dependent
    // VALU ops = ~150 cycles a row, not by the loads.)
    constexpr int PF = 1;
    U cur[PF * 8], nxt[PF * 8];
    const uint32_t Dv = D;
    auto fetch = [&](U* v, uint32_t b0) {
        const U* pb = p + (uint64_t)b0 * 8ull * Dv;
#pragma unroll
        for (int j = 0; j < PF; j++) {
            const bool in_range = b0 + j < nblocks;              // wave-uniform
#pragma unroll
            for (int i = 0; i < 8; i++) {
                v[j * 8 + i] = in_range ? *pb : (U)0;
                pb += Dv;
            }
        }
    };
    fetch(cur, 0);
    for (uint32_t b0 = 0; b0 < nblocks; b0 += PF) {
        fetch(nxt, b0 + PF);
#pragma unroll
        for (int j = 0; j < PF; j++) {
            if (b0 + j >= nblocks) break;
            int coef;
            if constexpr (W == 8) coef = (int)(int16_t)((ctr >> 5) << 4);          // :140-145
            else coef = (int)(int16_t)((uint32_t)(ctr >> 15) << 12);               // :224-232
            int grad = 0;
            U* const qb = q + (uint64_t)(b0 + j) * 8ull * Dv;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                int pred;
                if constexpr (W == 8) {
                    // byte 1 of the 16-bit product; the previous delta is taken unsigned in even columns (:163-168)
                    const int m = odd_col ? pd : (pd & 0xff);
                    pred = (int)(int8_t)(((uint32_t)(m * coef) >> 8) & 0xffu);
                } else {
                    pred = (int)(int16_t)((uint32_t)((pd * coef) >> 16) << 2);     // mulhi, << 2 (:240-242)
                }
                int delta, err;
                uint32_t x;
                if (DECODE) {
                    err = sx<W>((int)cur[j * 8 + i]);
                    delta = sx<W>(err + pred);
                    x = (pv + (uint32_t)delta) & ((1u << W) - 1u);
                    qb[(uint64_t)i * Dv] = (U)x;
                } else {
                    x = cur[j * 8 + i];
                    delta = sx<W>((int)(x - pv));
                    err = sx<W>(delta - pred);
                    qb[(uint64_t)i * Dv] = (U)err;
                }
                if (i & 1) grad = sx<W>(grad + (err > 0 ? pd : err < 0 ? sx<W>(-pd) : 0));   // _mm256_sign_epi8/16 (:173, :247)
                pv = x;
                pd = delta;
            }
            ctr += grad >> 2;                                                      // :195-206, :255-262
            if constexpr (W == 8) ctr = (int)(int16_t)ctr;
        }
#pragma unroll
        for (int k = 0; k < PF * 8; k++) cur[k] = nxt[k];
    }
    // the rows the vector code leaves alone: plain delta against the previous row (:266-273, :500)
    for (uint64_t i = (uint64_t)nblocks * 8ull * D + col; i < len; i += D) {
        const uint32_t v = in[i];
        if (DECODE) { pv = (pv + v) & ((1u << W) - 1u); out[i] = (U)pv; }
        else { out[i] = (U)(v - pv); pv = v; }
    }
}

// blocks the reference forecasts: rows / 8, less what its 32-byte stores would spill past the end (predict.cpp:96-103)
uint32_t xff_nblocks(uint64_t len, uint32_t D, int esz)
{
    const uint32_t V = 32u / (uint32_t)esz;
    const uint64_t blk = 8ull * D;
    int64_t nblocks = (int64_t)((len / D) / 8u);
    const uint32_t overrun = V - (D % V);
    if (overrun > len % blk) nblocks -= (int64_t)((overrun + blk - 1) / blk);
    return nblocks < 0 ? 0u : (uint32_t)nblocks;
}

template <typename U>
int xff_device(bool decode, const U* in, uint64_t len, uint32_t D, U* out, hipStream_t st)
{
    const uint32_t nb = xff_nblocks(len, D, (int)sizeof(U));
    const unsigned grid = (D + 63u) / 64u;
    if (decode) hipLaunchKernelGGL((xff_kernel<U, true>), dim3(grid), dim3(64), 0, st, in, len, D, nb, out);
    else hipLaunchKernelGGL((xff_kernel<U, false>), dim3(grid), dim3(64), 0, st, in, len, D, nb, out);
    return hipGetLastError() == hipSuccess ? 0 : fail(SPRINTZ_E_HIP, "xff transform launch");
}

int check(int kind, int esz, uint64_t len, uint16_t ndims)
{
    if (kind < 0 || kind > SPRINTZ_TRANSFORM_XFF) return fail(SPRINTZ_E_INVALID, "kind must be 0 (delta), 1 (double delta) or 2 (xff)");
    if (esz != 1 && esz != 2) return fail(SPRINTZ_E_INVALID, "elem_bytes must be 1 or 2");
    if (ndims == 0) return fail(SPRINTZ_E_INVALID, "ndims == 0");
    if (kind == SPRINTZ_TRANSFORM_XFF && len > 0xffffffffull) return fail(SPRINTZ_E_INVALID, "xff: len is 32 bits in the reference (predict.h:15)");
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

const char* sprintz_mi355x_transform_last_error(void) { return sprintz_mi355x_last_error(); }   // kept for ABI 1 callers

size_t sprintz_mi355x_transform_tmp_bytes(int kind, int elem_bytes, uint64_t len, uint16_t ndims)
{
    if (ndims == 0 || (elem_bytes != 1 && elem_bytes != 2)) return 0;
    if (kind == SPRINTZ_TRANSFORM_XFF) return 64;              // no scratch: a lane per column
    return make_plan(kind, elem_bytes, len, ndims).tmp_bytes + chain_tmp_bound(len * (uint64_t)elem_bytes) + 64;
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
    if (kind == SPRINTZ_TRANSFORM_XFF) {
        return elem_bytes == 1 ? xff_device<uint8_t>(false, (const uint8_t*)d_src, len, ndims, (uint8_t*)d_dest, st)
                               : xff_device<uint16_t>(false, (const uint16_t*)d_src, len, ndims, (uint16_t*)d_dest, st);
    }
    if (elem_bytes == 1) return kind ? encode_device<uint8_t, 1>((const uint8_t*)d_src, len, ndims, (uint8_t*)d_dest, st)
                                     : encode_device<uint8_t, 0>((const uint8_t*)d_src, len, ndims, (uint8_t*)d_dest, st);
    return kind ? encode_device<uint16_t, 1>((const uint16_t*)d_src, len, ndims, (uint16_t*)d_dest, st)
                : encode_device<uint16_t, 0>((const uint16_t*)d_src, len, ndims, (uint16_t*)d_dest, st);
}

#ifdef TR_CHAIN_TIMING
int sprintz_mi355x_dbg_chain_stamps(uint64_t* out, uint32_t ntiles) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chain_ts), (size_t)ntiles * 8 * sizeof(uint64_t)); }
#endif

int sprintz_mi355x_transform_decode_device(int kind, int elem_bytes, const void* d_src, uint64_t len, uint16_t ndims, void* d_dest,
                                           void* d_tmp, void* hip_stream)
{
    int rc = check(kind, elem_bytes, len, ndims);
    if (rc) return rc;
    if (!d_src || !d_dest || !d_tmp) return fail(SPRINTZ_E_INVALID, "null device pointer");
    if (!have_device()) return fail(SPRINTZ_E_NO_DEVICE, "no HIP device; there is no CPU fallback");
    if (len == 0) return 0;
    hipStream_t st = (hipStream_t)hip_stream;
    if (kind == SPRINTZ_TRANSFORM_XFF) {
        return elem_bytes == 1 ? xff_device<uint8_t>(true, (const uint8_t*)d_src, len, ndims, (uint8_t*)d_dest, st)
                               : xff_device<uint16_t>(true, (const uint16_t*)d_src, len, ndims, (uint16_t*)d_dest, st);
    }
    if (elem_bytes == 1) return kind ? decode_device<uint8_t, 1>((const uint8_t*)d_src, len, ndims, (uint8_t*)d_dest, (uint8_t*)d_tmp, st)
                                     : decode_device<uint8_t, 0>((const uint8_t*)d_src, len, ndims, (uint8_t*)d_dest, (uint8_t*)d_tmp, st);
    return kind ? decode_device<uint16_t, 1>((const uint16_t*)d_src, len, ndims, (uint16_t*)d_dest, (uint8_t*)d_tmp, st)
                : decode_device<uint16_t, 0>((const uint16_t*)d_src, len, ndims, (uint16_t*)d_dest, (uint8_t*)d_tmp, st);
}

// host forms with the reference's container: 6-byte header {u32 len; u16 ndims} (format.h:65-86)
int64_t sprintz_mi355x_transform_encode(int kind, int elem_bytes, const void* src, uint32_t len, void* dest, uint16_t ndims, int write_size)
{
    if (kind < 0 || kind > SPRINTZ_TRANSFORM_XFF) return fail(SPRINTZ_E_INVALID, "kind must be 0 (delta), 1 (double delta) or 2 (xff)");
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
    int rc = hipMemcpy(dx, src, nb, hipMemcpyHostToDevice) == hipSuccess ? 0 : fail(SPRINTZ_E_HIP, "hipMemcpy (host to device)");
    if (!rc) rc = sprintz_mi355x_transform_encode_device(kind, elem_bytes, dx, len, ndims, dy, nullptr);
    if (!rc && hipMemcpy(d, dy, nb, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(SPRINTZ_E_HIP, "hipMemcpy (device to host)");
    (void)hipFree(dx);
    (void)hipFree(dy);
    return rc ? rc : (int64_t)len + hdr;
}

// src carries the header unless len/ndims are given (the reference's 4-argument decode form, delta.h:19-21)
int64_t sprintz_mi355x_transform_decode(int kind, int elem_bytes, const void* src, void* dest, uint32_t raw_len, uint16_t raw_ndims)
{
    if (kind < 0 || kind > SPRINTZ_TRANSFORM_XFF) return fail(SPRINTZ_E_INVALID, "kind must be 0 (delta), 1 (double delta) or 2 (xff)");
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
    if (ndims == 0) return 0;                                  // delta.cpp:191, :637; predict.cpp:318
    if (len == 0) return 0;
    const size_t nb = (size_t)len * elem_bytes;
    const size_t tb = sprintz_mi355x_transform_tmp_bytes(kind, elem_bytes, len, ndims);
    void *dy = nullptr, *dx = nullptr, *dt = nullptr;
    if (hipMalloc(&dy, nb) != hipSuccess || hipMalloc(&dx, nb) != hipSuccess || hipMalloc(&dt, tb) != hipSuccess) {
        (void)hipFree(dy); (void)hipFree(dx);
        return fail(SPRINTZ_E_HIP, "hipMalloc");
    }
    int rc = hipMemcpy(dy, s, nb, hipMemcpyHostToDevice) == hipSuccess ? 0 : fail(SPRINTZ_E_HIP, "hipMemcpy (host to device)");
    if (!rc) rc = sprintz_mi355x_transform_decode_device(kind, elem_bytes, dy, len, ndims, dx, dt, nullptr);
    if (!rc && hipMemcpy(dest, dx, nb, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(SPRINTZ_E_HIP, "hipMemcpy (device to host)");
    (void)hipFree(dy);
    (void)hipFree(dx);
    (void)hipFree(dt);
    return rc ? rc : (int64_t)len;
}

}  // extern "C"

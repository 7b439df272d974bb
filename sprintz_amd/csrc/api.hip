// api.hip -- C-ABI of libsprintz_mi355x.so (include/sprintz_mi355x.h): argument
// checking, lane-mapping selection, kernel launches, the size-scan/compaction
// kernels and the host-pointer drop-in wrappers.  No codec arithmetic on the
// host; the only host-side stream logic is the framing walk that sizes the
// H2D copy of the length-less reference decompress() signature.
#include "../../include/sprintz_mi355x.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <ctime>
#include <sys/prctl.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "launch.h"
#include "encode_blk.h"
#include "decode_blk.h"
#include "decode_row.h"
#include "decode_lat.h"
#include "encode_lat.h"

using namespace sprintz;

namespace {

thread_local std::string g_last_error = "";

int fail(int code, const char* what, hipError_t e = hipSuccess)
{
    g_last_error = what;
    if (e != hipSuccess) {
        g_last_error += ": ";
        g_last_error += hipGetErrorString(e);
    }
    return code;
}

#define HIP_TRY(expr)                                              \
    do {                                                           \
        hipError_t e_ = (expr);                                    \
        if (e_ != hipSuccess) return fail(SPRINTZ_E_HIP, #expr, e_); \
    } while (0)

// Process-wide facts, established exactly once (the header promises re-entrancy from any thread):
// whether a HIP device exists at all, and the tuning knobs of the environment.
// (the knobs start from the environment, read once, and change only through sprintz_mi355x_set_option)
struct Process {
    bool have_device = false;
    std::atomic<int> no_fast{0};            // SPRINTZ_MI355X_NO_FAST: generic kernels only (A/B runs, tests)
    std::atomic<int> chunks_per_group{1};   // SPRINTZ_MI355X_CHUNKS_PER_GROUP (decode_fast read-ahead across chunks)
    std::atomic<int> dense_mode{1};         // SPRINTZ_MI355X_DENSE_MODE: how compress_batch_dense builds the container (see SPRINTZ_OPT_DENSE_MODE)
    std::atomic<int> enc_pair{1024};        // SPRINTZ_MI355X_ENC_PAIR: chunks from which row-major streams of 5 .. 64 columns are encoded with two columns per lane (0: never; see SPRINTZ_OPT_ENC_PAIR)
    std::atomic<int> blk_kernels{9};        // SPRINTZ_MI355X_BLK_KERNELS: which of round 6's delta kernels large batches take: bit 0 encode_blk (general layout), bit 1 decode_blk, bit 2 encode_blk_uni (univariate low-dim), bit 3 decode_row (wins over bit 1) on the shapes it wins on, bit 4 decode_row on every shape it fits
    std::atomic<int> blk_chunks{2049};      // SPRINTZ_MI355X_BLK_CHUNKS: batches of at least this many chunks take the block-parallel delta kernels (encode_blk.h; 0: never)
    std::atomic<int> lat_chunks{2048};      // SPRINTZ_MI355X_LAT_CHUNKS: batches of at most this many chunks decode with one workgroup per chunk (decode_lat.h; 0: never)
    std::atomic<int> ref_quirk{0};          // SPRINTZ_MI355X_REF_DECODER_QUIRK: decode as the reference DECODER does where it differs from the inverse of its encoder
    std::atomic<int> host_streams{4};       // SPRINTZ_MI355X_HOST_STREAMS: streams the host-pointer calls of all threads share per device (0: one per thread)
    std::atomic<int> host_wait{0};          // SPRINTZ_MI355X_HOST_WAIT: how a single call waits for its launches (see SPRINTZ_OPT_HOST_WAIT)
    std::atomic<int> split_lanes{1};        // SPRINTZ_MI355X_SPLIT_LANES: 8-bit streams of 65 .. 80 columns on 32 lanes x (pair + single) (see SPRINTZ_OPT_SPLIT_LANES)
};
Process& process()
{
    static Process p;
    static std::once_flag once;
    std::call_once(once, [] {
        int n = 0;
        p.have_device = hipGetDeviceCount(&n) == hipSuccess && n > 0;
        p.no_fast = getenv("SPRINTZ_MI355X_NO_FAST") != nullptr ? 1 : 0;
        if (const char* e = getenv("SPRINTZ_MI355X_DENSE_MODE")) {
            const int k = atoi(e);
            p.dense_mode = k <= 0 ? 0 : 1;
        }
        if (const char* e = getenv("SPRINTZ_MI355X_LAT_CHUNKS")) p.lat_chunks = atoi(e) < 0 ? 0 : atoi(e);
        if (const char* e = getenv("SPRINTZ_MI355X_BLK_CHUNKS")) p.blk_chunks = atoi(e) < 0 ? 0 : atoi(e);
        if (const char* e = getenv("SPRINTZ_MI355X_BLK_KERNELS")) p.blk_kernels = atoi(e) & 31;
        if (const char* e = getenv("SPRINTZ_MI355X_REF_DECODER_QUIRK")) p.ref_quirk = atoi(e) != 0 ? 1 : 0;
        if (const char* e = getenv("SPRINTZ_MI355X_HOST_STREAMS")) p.host_streams = atoi(e) < 0 ? 0 : (atoi(e) > 64 ? 64 : atoi(e));
        if (const char* e = getenv("SPRINTZ_MI355X_HOST_WAIT")) p.host_wait = atoi(e) < 0 || atoi(e) > 2 ? 0 : atoi(e);
        if (const char* e = getenv("SPRINTZ_MI355X_SPLIT_LANES")) p.split_lanes = atoi(e) != 0 ? 1 : 0;
        if (const char* e = getenv("SPRINTZ_MI355X_ENC_PAIR")) p.enc_pair = atoi(e) < 0 ? 0 : atoi(e);
        if (const char* e = getenv("SPRINTZ_MI355X_CHUNKS_PER_GROUP")) {
            const int k = atoi(e);
            p.chunks_per_group = k < 1 ? 1 : k > 64 ? 64 : k;
        }
    });
    return p;
}

int ensure_device()
{
    if (!process().have_device) return fail(SPRINTZ_E_NO_DEVICE, "no usable HIP device (libsprintz_mi355x has no CPU fallback)");
    return 0;
}

bool is_lowdim(int esz, int D) { return esz == 1 ? D <= 4 : D <= 2; }   // sprintz.cpp:34-50

struct Mapping { int log2DP; int cpl; };

// Choose lanes-per-chunk (DP = 2^k) and columns-per-lane so that DP*CPL >= D
// with little padding; among mappings within 75% of the best lane utilisation
// prefer the widest group (better coalescing of the D*esz-byte rows).
Mapping choose_mapping(int D, bool lowdim)
{
    if (lowdim) {
        int l = 0;
        while ((1 << l) < D) l++;
        return {l, 1};
    }
    double best = 0;
    for (int l = 0; l <= 6; l++)
        for (int c : kCplSet)
            if ((1 << l) * c >= D) best = std::max(best, (double)D / ((1 << l) * c));
    Mapping m{6, 8};
    bool found = false;
    for (int l = 6; l >= 0 && !found; l--) {
        for (int c : kCplSet) {
            if ((1 << l) * c < D) continue;
            if ((double)D / ((1 << l) * c) >= 0.75 * best) { m = {l, c}; found = true; break; }
        }
    }
    return m;
}

uint32_t next_pow2(uint32_t x)
{
    uint32_t p = 1;
    while (p < x) p <<= 1;
    return p;
}

size_t group_bytes_max(int esz, int D)
{
    const size_t hb = esz == 1 ? 3 : 4;
    return (2 * (size_t)D * hb + 7) / 8 + 16 * (size_t)D * esz;
}

// ---------------------------------------------------------------- compaction kernels

#ifndef SPRINTZ_BOUND_ALIGN
#define SPRINTZ_BOUND_ALIGN 128           // sprintz_mi355x_compress_bound is a multiple of this: slots start on 128-byte lines
#endif
#ifndef SPRINTZ_ENC_DRAIN_ALIGN
#define SPRINTZ_ENC_DRAIN_ALIGN 128       // encode_fast.h / encode_wide.h: granularity of the window's flushes to the slot
#endif
constexpr int kScanBlock = 1024;

__global__ void __launch_bounds__(kScanBlock) scan_local_kernel(const uint32_t* sizes, uint64_t n, uint32_t align,
                                                                uint64_t* offsets, uint64_t* block_sums)
{
    __shared__ uint64_t sh[kScanBlock];
    const uint64_t i = (uint64_t)blockIdx.x * kScanBlock + threadIdx.x;
    const uint64_t a = align - 1;
    uint64_t v = i < n ? (((uint64_t)sizes[i] + a) & ~a) : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < kScanBlock; off <<= 1) {
        uint64_t t = threadIdx.x >= (unsigned)off ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    if (i < n) offsets[i] = sh[threadIdx.x] - v;
    if (threadIdx.x == kScanBlock - 1) block_sums[blockIdx.x] = sh[threadIdx.x];
}

__global__ void __launch_bounds__(kScanBlock) scan_blocks_kernel(uint64_t* block_sums, uint64_t nblocks, uint64_t* total_out)
{
    __shared__ uint64_t sh[kScanBlock];
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint64_t base = 0; base < nblocks; base += kScanBlock) {
        const uint64_t i = base + threadIdx.x;
        const uint64_t v = i < nblocks ? block_sums[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < kScanBlock; off <<= 1) {
            uint64_t t = threadIdx.x >= (unsigned)off ? sh[threadIdx.x - off] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nblocks) block_sums[i] = carry + sh[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == kScanBlock - 1) carry += sh[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = carry;
}

__global__ void __launch_bounds__(kScanBlock) scan_add_kernel(uint64_t* offsets, uint64_t n, const uint64_t* block_sums)
{
    const uint64_t i = (uint64_t)blockIdx.x * kScanBlock + threadIdx.x;
    if (i < n) offsets[i] += block_sums[blockIdx.x];
}

// batches of up to kScanOne chunks: the whole scan in ONE workgroup (a thread takes kScanPer consecutive sizes, the 1 024 partial
// sums are scanned in LDS) -- one launch instead of three where the launches are what the scan costs (BASELINE config 5: 6 554 chunks)
constexpr int kScanPer = 16, kScanOne = kScanBlock * kScanPer;
__global__ void __launch_bounds__(kScanBlock) scan_one_kernel(const uint32_t* sizes, uint64_t n, uint32_t align, uint64_t* offsets)
{
    __shared__ uint64_t sh[kScanBlock];
    const uint64_t a = align - 1, i0 = (uint64_t)threadIdx.x * kScanPer;
    uint64_t v[kScanPer], sum = 0;
#pragma unroll
    for (int k = 0; k < kScanPer; k++) {
        v[k] = i0 + k < n ? (((uint64_t)sizes[i0 + k] + a) & ~a) : 0;
        sum += v[k];
    }
    // inclusive scan of the 1 024 partial sums: inside a wavefront with shuffles, over the 16 wavefronts through LDS (round 5: ten
    // Hillis-Steele rounds of two barriers each were most of this one-workgroup kernel's 9 - 12 us)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint64_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)incl, off, 64), hi = (uint32_t)__shfl_up((int)(uint32_t)(incl >> 32), off, 64);
        if ((int)lane >= off) incl += ((uint64_t)hi << 32) | lo;
    }
    if (lane == 63) sh[wave] = incl;
    __syncthreads();
    uint64_t wbase = 0, total = 0;
#pragma unroll
    for (uint32_t k = 0; k < kScanBlock / 64; k++) { wbase += k < wave ? sh[k] : 0ull; total += sh[k]; }
    uint64_t run = wbase + incl - sum;
#pragma unroll
    for (int k = 0; k < kScanPer; k++) {
        if (i0 + k < n) offsets[i0 + k] = run;
        run += v[k];
    }
    if (threadIdx.x == kScanBlock - 1) offsets[n] = total;
}

#ifndef SPRINTZ_COPY_SMALL_LOG2
#define SPRINTZ_COPY_SMALL_LOG2 5
#endif
// one wavefront per chunk: slot -> dense (LOG2L = 5: half a wavefront per chunk -- slots of at most 2 KB, where a chunk's stream is a few
// hundred bytes and 64 lanes x 16 bytes leave most of the wavefront idle: BASELINE config 1's 440-byte streams, compress 0.434 -> 0.402 ms; a quarter: 0.403)
template <int LOG2L>
__global__ void __launch_bounds__(kThreads) compact_copy_kernel(const uint8_t* slots, uint64_t slot_stride, const uint32_t* sizes,
                                                                const uint64_t* offsets, uint64_t nchunks, uint32_t align,
                                                                uint8_t* dense)
{
    const uint64_t c = ((uint64_t)blockIdx.x * kThreads + threadIdx.x) >> LOG2L;
    const uint32_t lane = threadIdx.x & ((1u << LOG2L) - 1u);
    if (c >= nchunks) return;
    const uint8_t* s = slots + c * slot_stride;
    uint8_t* d = dense + offsets[c];
    const uint32_t sz = sizes[c];
    if (align == 16) {
        const uint32_t nunits = (sz + 15u) >> 4;     // slots are zero padded to 16
        sprintz::copy_verbatim<false>(s, d, nunits << 4, lane, 1u << LOG2L);      // (four 16-byte loads a lane in flight before its first store)
    } else {
        for (uint32_t j = lane; j < sz; j += 1u << LOG2L) d[j] = s[j];
    }
}

// Chunks too short for one stream group (n < 128 or n < 16 * ndims in the general layout: BASELINE config 3 at 1 KB chunks) are
// their 8-byte header + the samples themselves (sprintz_xff_rle.cpp:116-124, :158-160; encode_kernel.h writes the same bytes
// into a slot).  Every size is known before the launch, so the 16-byte aligned container needs no scan and no slot: one
// wavefront per chunk copies the samples to where they end up -- one pass over the data instead of two.
__global__ void __launch_bounds__(kThreads) verbatim_dense_kernel(const uint8_t* src, uint64_t total_len, uint32_t chunk_len, uint32_t esz, uint32_t D,
                                                                  uint64_t nchunks, uint8_t* dense, uint64_t* offsets, uint32_t* sizes, int64_t* rets)
{
    const uint64_t c = ((uint64_t)blockIdx.x * kThreads + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    if (c >= nchunks) return;
    const uint64_t first = c * (uint64_t)chunk_len;
    const uint32_t n = (uint32_t)(total_len - first < chunk_len ? total_len - first : chunk_len);
    const uint64_t stride = ((uint64_t)8 + (uint64_t)chunk_len * esz + 15u) & ~(uint64_t)15;     // every chunk but the last is chunk_len long
    const uint32_t size = 8u + n * esz, asize = (size + 15u) & ~15u;
    uint8_t* d = dense + c * stride;
    sprintz::copy_verbatim<true>(src + first * esz, d + 8, n * esz, lane, 64u);
    for (uint32_t j = size + lane; j < asize; j += 64u) d[j] = 0;                                 // the container's alignment padding is zeros
    if (lane == 0) {
        ((uint32_t*)d)[0] = 0;                                                                    // no groups (format.h:36-45)
        ((uint32_t*)d)[1] = (n & 0xffffu) | (D << 16);
        sizes[c] = size;
        if (rets) rets[c] = (int64_t)(size / esz);
        offsets[c] = c * stride;
        if (c == nchunks - 1) offsets[nchunks] = c * stride + asize;
    }
}

// The way back for such batches (chunk_len < 128 or < 16 * ndims: no stream of a valid batch holds a group): one wavefront per
// chunk checks the 8-byte header (no groups, ndims, the tail inside the stream and inside the chunk) and copies the samples --
// instead of decode_kernel.h's whole state machine around the same copy.  A stream that does announce groups cannot be valid
// at this chunk length (one group is 16 * ndims samples) and is SPRINTZ_E_CORRUPT.
__global__ void __launch_bounds__(kThreads) verbatim_decode_kernel(const uint8_t* comp, const uint64_t* offsets, uint64_t nchunks, uint32_t chunk_len,
                                                                   uint32_t esz, uint32_t D, uint8_t* out, int64_t* rets)
{
    const uint64_t c = ((uint64_t)blockIdx.x * kThreads + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    if (c >= nchunks) return;
    const uint64_t off = offsets[c], slen = offsets[c + 1] - off;
    const uint8_t* s = comp + off;
    bool bad = slen < 8;
    uint32_t remaining = 0;
    if (!bad) {
        const uint32_t w0 = sprintz::load_u32_any(s), w1 = sprintz::load_u32_any(s + 4);
        remaining = w1 & 0xffffu;
        bad = w0 != 0 || (w1 >> 16) != D || remaining > chunk_len || (uint64_t)remaining * esz > slen - 8;
    }
    if (!bad) sprintz::copy_verbatim<true>(s + 8, out + c * (uint64_t)chunk_len * esz, remaining * esz, lane, 64u);
    if (lane == 0 && rets) rets[c] = bad ? sprintz::kErrCorrupt : (int64_t)remaining;
}

}  // namespace

namespace sprintz {
int set_error(int code, const char* what) { return fail(code, what); }
// exclusive scan of (aligned) u32 sizes into u64 offsets[n+1]; tmp = sprintz_mi355x_compact_tmp_bytes(n)
hipError_t launch_size_scan(const uint32_t* d_sizes, uint64_t n, uint32_t align, uint64_t* d_offsets, void* d_tmp, hipStream_t st)
{
    if (n == 0) return hipMemsetAsync(d_offsets, 0, 8, st);
    if (n <= (uint64_t)kScanOne) {
        hipLaunchKernelGGL(scan_one_kernel, dim3(1), dim3(kScanBlock), 0, st, d_sizes, n, align, d_offsets);
        return hipGetLastError();
    }
    const uint64_t nblocks = (n + kScanBlock - 1) / kScanBlock;
    uint64_t* tmp = (uint64_t*)d_tmp;
    hipLaunchKernelGGL(scan_local_kernel, dim3((unsigned)nblocks), dim3(kScanBlock), 0, st, d_sizes, n, align, d_offsets, tmp);
    hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(kScanBlock), 0, st, tmp, nblocks, d_offsets + n);
    hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)nblocks), dim3(kScanBlock), 0, st, d_offsets, n, tmp);
    return hipGetLastError();
}
}  // namespace sprintz

namespace {

// ---------------------------------------------------------------- launch helpers

int check_common(int codec, int esz, uint16_t ndims)
{
    if (codec < SPRINTZ_CODEC_DELTA || codec > SPRINTZ_CODEC_XFF_NORLE)
        return fail(SPRINTZ_E_INVALID, "codec must be 0 (delta), 1 (xff), 2 (delta, no RLE), 3 (bit-packing only) or 4 (xff, no RLE)");
    if (codec == SPRINTZ_CODEC_XFF_NORLE && esz != 1) return fail(SPRINTZ_E_UNSUPPORTED, "the non-RLE xff codec exists for 8-bit elements only");
    if (esz != 1 && esz != 2) return fail(SPRINTZ_E_INVALID, "elem_bytes must be 1 or 2");
    if (ndims == 0) return fail(SPRINTZ_E_INVALID, "ndims == 0 (reference: sprintz.cpp:36 returns -1)");
    if (ndims > SPRINTZ_MI355X_MAX_NDIMS) return fail(SPRINTZ_E_UNSUPPORTED, "ndims above SPRINTZ_MI355X_MAX_NDIMS");
    return 0;
}

// The stream header's remaining_len is a uint16 (format.h:40): from 4 096 columns on a chunk's verbatim tail -- the whole chunk when
// it is shorter than one group, otherwise at most two blocks (the "<" codecs: sprintz_delta_rle.cpp:226) or one block and the
// ragged rest -- can exceed 65 535 elements.  The reference's single call then writes a header that decodes to a prefix, and the
// drop-in symbols reproduce that; a BATCH that silently loses samples is not acceptable, so the batched entry points refuse it.
int check_batch_tail(uint64_t total_len, uint32_t chunk_len, uint16_t ndims)
{
    if (ndims < 4096 || chunk_len == 0) return 0;
    auto tail_max = [&](uint64_t n) -> uint64_t {
        const uint64_t blk = 8ull * ndims;
        if (n < 128 || n < 2 * blk) return n;
        return n % blk == 0 ? 2 * blk : blk + n % blk;
    };
    const uint64_t last = total_len % chunk_len;
    const bool full = total_len >= chunk_len;
    if ((full && tail_max(chunk_len) > 0xffffu) || (last && tail_max(last) > 0xffffu))
        return fail(SPRINTZ_E_UNSUPPORTED, "a chunk's verbatim tail can exceed the stream header's 16-bit remaining_len at this ndims x chunk_len: the batch would decode to a prefix");
    return 0;
}

// more than 2 047 columns: the column-tiled kernels build the stream with device-scope atomics on the slot and read their own output back
// (any_ndims.hip, "big"): that needs ordinary device memory -- a mapped host or managed buffer is refused instead of producing a damaged stream
bool is_plain_device_memory(const void* p)
{
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeDevice;
}

// query-on-compressed options of one decode launch (decode_kernel.h: Q template parameter)
// a single call served straight from the caller thread's mapped host buffer by ONE launch of a workgroup-per-chunk kernel
// (decode_lat.h / encode_lat.h): no staging kernel in front, no runtime wait behind -- the kernel's last store is the
// call's ticket into a mapped host word
struct HostCall {
    uint64_t off0 = 0, off1 = 0;   // decode: the stream is comp[off0, off1)
    uint64_t* flag = nullptr;      // device view of the word
    uint64_t ticket = 0;
};

struct QuerySpec {
    int q = kQueryOff;          // kQueryOff / kQueryMaterialize / kQueryReduceOnly
    int qop = 0;                // 1 max, 2 sum
    uint64_t* qres = nullptr;   // [nchunks][ndims]
    int general = 0;            // 1: general row-major layout for every ndims (the reference's *_rowmajor_*_rle_* family)
    uint64_t col_stride = 0;    // != 0: column-major destination (DecodeArgs::col_stride)
    const HostCall* hc = nullptr;
};

// One chunk's working set of the workgroup-per-chunk kernels (decode_lat.h / encode_lat.h) must fit a workgroup's LDS: up to 16 KB of
// samples several workgroups share a CU (what the batch limits of SPRINTZ_OPT_LAT_CHUNKS were measured with); larger chunks -- up to
// ~40 KB of uint16, ~24 KB of uint8: 150 KB of LDS, a workgroup a CU -- only for batches that leave most CUs empty anyway (single calls)
bool lat_chunk_fits(bool encode, int esz, uint64_t nchunks, uint32_t chunk_len, int D)
{
    const uint64_t bytes = (uint64_t)chunk_len * esz;
    if (bytes > (48u << 10) || (bytes > kLatMaxChunkBytes && nchunks > 64)) return false;
    // the carve and the 16-bit position limit are checked for EVERY size: a shape whose working set does not fit goes to the
    // lane-per-column kernels instead of failing its launch
    const uint32_t bound = (uint32_t)sprintz_mi355x_compress_bound(esz, chunk_len, (uint16_t)D);
    if (bound > 60000u) return false;                            // (stream positions travel in 16 bits between the kernels' phases)
    const uint32_t total = encode ? enc_lat_carve(bound, chunk_len, (uint32_t)D, (uint32_t)esz).total : lat_carve(bound, chunk_len, (uint32_t)D).total;
    return total <= 150u * 1024u;
}

bool decode_ref_quirk(int codec, int esz, bool lowdim)
{
    // (only 16-bit general-layout FIRE streams have the divergence: sprintz_xff_rle.cpp:893-901)
    return esz == 2 && codec == SPRINTZ_CODEC_XFF && !lowdim && process().ref_quirk.load(std::memory_order_relaxed);
}

// small batches: one WORKGROUP per chunk (decode_lat.h) -- a chunk's 40 dependent group steps on one lane group take 50 us
// however few chunks there are; split into a header walk, parallel bit extraction, the bare recurrence and a prefix sum it is ~13
bool decode_lat_fits(int codec, int esz, uint64_t nchunks, uint32_t chunk_len, int D, int noheader, const QuerySpec& qs, const void* d_out)
{
    const bool norle = codec >= SPRINTZ_CODEC_DELTA_NORLE;
    const bool lowdim = (qs.general || norle) ? false : is_lowdim(esz, D);
    return !norle && !noheader && !qs.col_stride && !decode_ref_quirk(codec, esz, lowdim) && qs.q == kQueryOff && D <= 64 &&
           lat_chunk_fits(false, esz, nchunks, chunk_len, D) && chunk_len >= 16u * (uint32_t)D && ((uintptr_t)d_out % 16) == 0 &&
           (nchunks == 1 || ((uint64_t)chunk_len * esz) % 16 == 0) &&      // (a chunk's output starts 16-byte aligned; its end may lie anywhere)
           // (about one round of workgroups on the chip is where it wins: 5 a CU at 8 columns -- measured 33 vs 47 us at 1 250 chunks, 41 vs 47
           //  at 2 048, 59 vs 47 at 3 072; with more columns a chunk has fewer groups to walk and the lane-per-column kernel catches up
           //  sooner: 32 columns 11.6 vs 14.7 at 640 chunks, 19.7 vs 14.8 at 1 250 -- a third of the limit from 17 columns on)
           nchunks <= (uint64_t)process().lat_chunks.load(std::memory_order_relaxed) / (D > 16 ? 3u : 1u) && !process().no_fast.load(std::memory_order_relaxed);
}

int decode_launch(int codec, int esz, const void* d_comp, const uint64_t* d_offsets, uint64_t nchunks,
                  uint32_t chunk_len, uint16_t ndims, void* d_out, int64_t* d_rets, hipStream_t st,
                  int noheader, uint32_t nh_ngroups, uint32_t nh_remaining, const QuerySpec& qs = QuerySpec{})
{
    if (nchunks == 0) return 0;
    const int D = ndims;
    const bool norle = codec >= SPRINTZ_CODEC_DELTA_NORLE;      // general layout for every ndims, generic kernels
    const bool lowdim = (qs.general || norle) ? false : is_lowdim(esz, D);
    const Mapping m = choose_mapping(D, lowdim);
    const int DP = 1 << m.log2DP;

    DecodeArgs a{};
    a.comp = (const uint8_t*)d_comp;
    a.offsets = d_offsets;
    a.nchunks = nchunks;
    a.chunk_len = chunk_len;
    a.D = D;
    a.log2DP = m.log2DP;
    a.out = d_out;
    a.rets = d_rets;
    a.noheader = noheader;
    a.nh_ngroups = nh_ngroups;
    a.nh_remaining = nh_remaining;
    a.chunks_per_group = 1;
    a.qop = qs.qop;
    a.qres = qs.qres;
    a.norle = norle ? (codec == SPRINTZ_CODEC_XFF_NORLE ? 2 : 1) : 0;
    a.raw = codec == SPRINTZ_CODEC_BITPACK_NORLE ? 1 : 0;
    a.col_stride = qs.col_stride;
    const uint64_t cs = qs.col_stride;
    a.quirk = decode_ref_quirk(codec, esz, lowdim) ? 1 : 0;
    if (qs.hc && !decode_lat_fits(codec, esz, nchunks, chunk_len, D, noheader, qs, d_out)) return fail(SPRINTZ_E_HIP, "internal: host call on a kernel that cannot end it");

    // 513 .. 2047 columns: one workgroup per chunk (any_ndims.hip) -- the RLE codecs, row-major, plain decode
    if (D > 512) {
        if (norle || cs || qs.q != kQueryOff) return fail(SPRINTZ_E_UNSUPPORTED, "more than 512 columns: the RLE codecs, row-major, without query only");
        if (nchunks > 0x7fffffffull) return fail(SPRINTZ_E_INVALID, "too many chunks for one launch");
        if (D > 2047) {                                        // column tiles; the FIRE counters in stream-ordered scratch (any_ndims.hip, "big")
            if (!is_plain_device_memory(d_out)) return fail(SPRINTZ_E_INVALID, "more than 2047 columns: the output must be device memory (hipMalloc), not mapped host or managed memory");
            int32_t* counters = nullptr;
            const bool fire = codec == SPRINTZ_CODEC_XFF;
            if (fire && (uint64_t)nchunks * (uint64_t)D * 4 > (1ull << 30)) return fail(SPRINTZ_E_UNSUPPORTED, "more than 2047 columns, FIRE: the counters' scratch (nchunks x ndims x 4 bytes) is limited to 1 GiB a launch: split the batch");
            if (fire && hipMallocAsync((void**)&counters, (size_t)nchunks * (size_t)D * 4, st) != hipSuccess) return fail(SPRINTZ_E_HIP, "hipMallocAsync of the counters' scratch (not available during stream capture)");
            const hipError_t eb = launch_decode_big(8 * esz, fire, (unsigned)nchunks, st, a, counters);
            if (counters) (void)hipFreeAsync(counters, st);
            if (eb != hipSuccess) return fail(SPRINTZ_E_HIP, "decode_big kernel launch", eb);
            return 0;
        }
        const hipError_t ea = launch_decode_any(8 * esz, codec == SPRINTZ_CODEC_XFF, (unsigned)nchunks, st, a);
        if (ea != hipSuccess) return fail(SPRINTZ_E_HIP, "decode_any kernel launch", ea);
        return 0;
    }

    // batches whose chunks are too short for a stream group: header check + copy (verbatim_decode_kernel)
    // (only where a chunk cannot hold a group at all, chunk_len < 16 D: a stream of 16 D <= chunk_len < 128 elements that announces
    //  groups is one the reference ENCODER never writes but its decoder reads -- that one goes to the decoders below)
    if (!norle && !lowdim && !noheader && !cs && qs.q == kQueryOff && chunk_len < 16u * (uint32_t)D &&
        !process().no_fast.load(std::memory_order_relaxed)) {
        const uint64_t vgrid = (nchunks * 64 + kThreads - 1) / kThreads;
        if (vgrid > 0x7fffffffull) return fail(SPRINTZ_E_INVALID, "too many chunks for one launch");
        hipLaunchKernelGGL(verbatim_decode_kernel, dim3((unsigned)vgrid), dim3(kThreads), 0, st, (const uint8_t*)d_comp, d_offsets, nchunks, chunk_len,
                           (uint32_t)esz, (uint32_t)D, (uint8_t*)d_out, d_rets);
        HIP_TRY(hipGetLastError());
        return 0;
    }

    // LDS-transposed 16-byte stores need every 8 x D block of the output 16-byte aligned
    const size_t blk_bytes = (size_t)8 * D * esz;
    const size_t stride = ((blk_bytes + 15) & ~(size_t)15) + 16;     // +16: spread groups over LDS banks
    const size_t groups_per_block = kThreads / DP;
    size_t shmem = 0;
    a.vec_store = 0;
    if (!cs && blk_bytes % 16 == 0 && (qs.q == kQueryReduceOnly || ((uintptr_t)d_out % 16) == 0) && ((uint64_t)chunk_len * esz) % 16 == 0 &&
        stride * groups_per_block <= 64 * 1024) {
        a.vec_store = 1;
        a.lds_group_stride = (uint32_t)stride;
        shmem = stride * groups_per_block;
    }

    // Fast path (decode_fast.h): general layout, one column per lane, headered stream,
    // vector stores legal, and the power-of-two group at least half full.
    int fdp = 4, fcpl = 1;
    while (fdp < D && fdp < 64) fdp <<= 1;
    while (fdp * fcpl < D) fcpl <<= 1;                         // 2 / 4 columns per lane for D in 65..256
    // (for 65..96 columns <DP 32, CPL 3> keeps 84 % of the lanes busy instead of 62 % but holds 9 waves per CU
    //  instead of 12: measured slower, u8 D=80 1.29 -> 1.26 TB/s, u16 D=80 1.63 -> 1.34)
    // (two columns per lane at D = 8, i.e. <DP 4, CPL 2>, halves the lanes per chunk but not the LDS per chunk:
    //  8 waves per CU instead of 16, measured 0.494 vs 0.400 ms -- the doubled ILP does not replace the lost waves)
    // (32-bit offsets inside one wavefront's span of the output)
    // and chunks not much shorter than the read-ahead ring (it is filled before the first header is parsed)
    // 8 bits, 65 .. 80 columns, plain row-major decode: 32 lanes x (a pair + a single column), two chunks a wavefront, the LDS
    // carve sized for 80 columns so that 12 wavefronts a CU stay resident (decode_fast.h, SPLIT)
    int fds = 0;
    if (esz == 1 && D > 64 && D <= 80 && !cs && qs.q == kQueryOff && process().split_lanes.load(std::memory_order_relaxed)) { fdp = 32; fcpl = 3; fds = 80; }
    // 16 bits, the same widths: 64 x 2 stays, with the carve of 80 columns (12.2 KB a chunk instead of 17.8: 12 waves a CU instead of 8)
    if (esz == 2 && D > 64 && D <= 80 && !cs && qs.q == kQueryOff && process().split_lanes.load(std::memory_order_relaxed)) fds = 80;
    const size_t fring = decode_fast_lds_bytes(8 * esz, fdp, fcpl, D, cs != 0 && fcpl == 1, fds);
    const bool fast_common = !lowdim && !a.raw && !noheader && D <= 256 && 2 * D > fdp * fcpl && (uint64_t)chunk_len * esz * 2 >= fring &&
                             !process().no_fast.load(std::memory_order_relaxed);
    // column-major: a lane's 8 samples per block are one aligned 16-byte (8-byte) piece of its column
    const bool fast = cs ? fast_common && qs.q == kQueryOff && cs % 8 == 0 && (chunk_len / (uint32_t)D) % 8 == 0 &&
                               ((uintptr_t)d_out % 16) == 0 && (uint64_t)D * cs * esz < 0xf0000000ull
                         : fast_common && a.vec_store && (uint64_t)chunk_len * esz * 64 * 64 < 0xf0000000ull;
    hipError_t e;
    if (decode_lat_fits(codec, esz, nchunks, chunk_len, D, noheader, qs, d_out)) {     // (a.raw is a run-less codec)
        if (qs.hc) { a.offsets = nullptr; a.one_off0 = qs.hc->off0; a.one_off1 = qs.hc->off1; a.host_flag = qs.hc->flag; a.host_ticket = qs.hc->ticket; }
        int ldp = 4;
        while (ldp < D) ldp <<= 1;
        if (esz == 1 && ldp < 8 && !lowdim) ldp = 8;
        e = launch_decode_lat(8 * esz, codec == SPRINTZ_CODEC_XFF, ldp, lowdim, (unsigned)nchunks, (uint32_t)sprintz_mi355x_compress_bound(esz, chunk_len, ndims), st, a);
        if (e != hipSuccess) return fail(SPRINTZ_E_HIP, "decode_lat kernel launch", e);
        return 0;
    }
    // large batches of the DELTA codec, general layout, rows of whole dwords: a lane per dword-wide column group, blocks in order (decode_row.h)
    {
        const int blk_from = process().blk_chunks.load(std::memory_order_relaxed);
        if (blk_from > 0 && (process().blk_kernels.load(std::memory_order_relaxed) & 8) && nchunks >= (uint64_t)blk_from && codec == SPRINTZ_CODEC_DELTA && !lowdim && !noheader && !cs &&
            qs.q == kQueryOff && !qs.hc && ((uintptr_t)d_out % 4) == 0 && ((uintptr_t)d_comp % 4) == 0 && !process().no_fast.load(std::memory_order_relaxed)) {
            const RowDecGeom g = row_dec_geom((uint32_t)esz, chunk_len, (uint32_t)D);
            // where it wins (tools/blk_shapes.py, profiles/r6_blk_shapes.txt; 10 KB chunks, ms against the lane-per-column kernels): 8-bit rows of 32 / 48 / 64 / 80 /
            // 128 / 256 columns 0.153 / 0.202 / 0.121 / 0.132 / 0.136 / 0.187 against 0.172 / 0.237 / 0.184 / 0.172 / 0.155 / 0.539; where it does not: 16 8-bit columns
            // (4 lanes a chunk) 0.233 against 0.182, and 16-bit elements -- two fields a dword carry the same per-row work as four -- 8 / 16 / 24 / 128 columns 0.129 /
            // 0.103 / 0.171 / 0.186 against 0.114 / 0.096 / 0.136 / 0.132 (32 and 64 columns level).  Mask bit 4 takes every shape the kernel fits (tests).
            const bool wins = (esz == 1 && g.U >= 8u) || (process().blk_kernels.load(std::memory_order_relaxed) & 16);
            // (32-bit offsets inside the kernel: the output and -- whatever the streams' lengths -- the container below 4 GB)
            const bool below_4g = (uint64_t)nchunks * chunk_len * esz < 0xf0000000ull && (uint64_t)nchunks * sprintz_mi355x_compress_bound(esz, chunk_len, ndims) < 0xf0000000ull;
            if (g.ok && below_4g && wins) {
                const uint64_t rgrid = (nchunks + 4ull * g.G - 1) / (4ull * g.G);
                if (rgrid > 0x7fffffffull) return fail(SPRINTZ_E_INVALID, "too many chunks for one launch");
                e = launch_decode_row(8 * esz, (unsigned)rgrid, st, a, g);
                if (e != hipSuccess) return fail(SPRINTZ_E_HIP, "decode_row kernel launch", e);
                return 0;
            }
        }
    }
    // ... or the block-parallel decoder (decode_blk.h)
    {
        const int blk_from = process().blk_chunks.load(std::memory_order_relaxed);
        if (blk_from > 0 && (process().blk_kernels.load(std::memory_order_relaxed) & 2) && nchunks >= (uint64_t)blk_from && codec == SPRINTZ_CODEC_DELTA && !lowdim && !noheader && !cs && qs.q == kQueryOff && !qs.hc &&
            ((uintptr_t)d_out % 16) == 0 && !process().no_fast.load(std::memory_order_relaxed)) {
            const BlkDecGeom g = blk_dec_geom((uint32_t)esz, chunk_len, (uint32_t)D, (uint32_t)sprintz_mi355x_compress_bound(esz, chunk_len, ndims));
            if (g.ok) {
                const uint64_t bgrid = (nchunks + g.CPW - 1) / g.CPW;
                if (bgrid > 0x7fffffffull) return fail(SPRINTZ_E_INVALID, "too many chunks for one launch");
                e = launch_decode_blk(8 * esz, (unsigned)bgrid, st, a, g);
                if (e != hipSuccess) return fail(SPRINTZ_E_HIP, "decode_blk kernel launch", e);
                return 0;
            }
        }
    }
    if (fast) {
        a.log2DP = 0;
        while ((1 << a.log2DP) < fdp) a.log2DP++;
        const size_t fgroups = kThreads / fdp;
        // (padding the stride by 16 / 32 / 48 bytes to move the groups' staging rows onto other banks: no change, 0.4225 ms each)
        const size_t fstride = decode_fast_lds_bytes(8 * esz, fdp, fcpl, D, cs != 0 && fcpl == 1, fds);
        a.lds_group_stride = (uint32_t)fstride;
        // consecutive chunks per lane group.  Measured on MI355X (cfg2, 131072 chunks): k = 1 / 2 / 4 /
        // 8 -> 0.498 / 0.496 / 0.510 / 0.560 ms: one generation of lock-stepped groups is no faster
        // than four staggered ones, so the default stays at one chunk per group (env knob for tuning).
        a.chunks_per_group = (uint32_t)process().chunks_per_group.load(std::memory_order_relaxed);
        const uint64_t ngroups_launch = (nchunks + a.chunks_per_group - 1) / a.chunks_per_group;
        const uint64_t fthreads = ngroups_launch * (uint64_t)fdp;
        const uint64_t fgrid = (fthreads + kThreads - 1) / kThreads;
        if (fgrid > 0x7fffffffull) return fail(SPRINTZ_E_INVALID, "too many chunks for one launch");
        e = esz == 1 ? launch_decode_fast_w8((codec == SPRINTZ_CODEC_XFF || codec == SPRINTZ_CODEC_XFF_NORLE), fdp, fcpl, D == fdp * fcpl, qs.q, fds, (unsigned)fgrid, fstride * fgroups, st, a)
                     : launch_decode_fast_w16((codec == SPRINTZ_CODEC_XFF || codec == SPRINTZ_CODEC_XFF_NORLE), fdp, fcpl, D == fdp * fcpl, qs.q, fds, (unsigned)fgrid, fstride * fgroups, st, a);
        if (e != hipSuccess) return fail(SPRINTZ_E_HIP, "decode_fast kernel launch", e);
        return 0;
    }
    // univariate streams: one lane per chunk, LDS ring in, quad-transposed 64-byte bursts out (decode_uni.h)
    // (and the other low-dim shapes: 2 columns, 3 and 4 at 8 bits)
    if (lowdim && (D <= 2 || esz == 1) && !noheader && !cs && !process().no_fast.load(std::memory_order_relaxed)) {
        const uint64_t ugrid = (nchunks + 255) / 256;
        if (ugrid > 0x7fffffffull) return fail(SPRINTZ_E_INVALID, "too many chunks for one launch");
        e = esz == 1 ? launch_decode_uni_w8(codec == SPRINTZ_CODEC_XFF, D, qs.q, (unsigned)ugrid, st, a)
                     : launch_decode_uni_w16(codec == SPRINTZ_CODEC_XFF, D, qs.q, (unsigned)ugrid, st, a);
        if (e != hipSuccess) return fail(SPRINTZ_E_HIP, "decode_uni kernel launch", e);
        return 0;
    }
    const uint64_t threads = nchunks * (uint64_t)DP;
    const uint64_t grid = (threads + kThreads - 1) / kThreads;
    if (grid > 0x7fffffffull) return fail(SPRINTZ_E_INVALID, "too many chunks for one launch");
    e = esz == 1 ? launch_decode_w8((codec == SPRINTZ_CODEC_XFF || codec == SPRINTZ_CODEC_XFF_NORLE), lowdim, m.cpl, qs.q, (unsigned)grid, shmem, st, a)
                 : launch_decode_w16((codec == SPRINTZ_CODEC_XFF || codec == SPRINTZ_CODEC_XFF_NORLE), lowdim, m.cpl, qs.q, (unsigned)grid, shmem, st, a);
    if (e != hipSuccess) return fail(SPRINTZ_E_HIP, "decode kernel launch", e);
    return 0;
}

// what the encode launch needs to build the dense container itself (compact_tail.h); `fused` reports whether it did
struct DenseRequest {
    void* d_dense = nullptr;
    uint64_t* d_offsets = nullptr;
    void* d_tmp = nullptr;
    bool fused = false;
};

// small batches: one WORKGROUP per chunk (encode_lat.h), the counterpart of decode_lat.h -- 90 us for ONE 10 KB chunk on a lane
// group, ~20 with the coefficient chain and the RLE state machine as the only serial parts (the container, if one was asked
// for, is then built by the scan + copy passes: dense->fused stays false)
bool encode_lat_fits(int codec, int esz, uint64_t nchunks, uint32_t chunk_len, int D, uint64_t col_stride, const void* d_src, const void* d_slots, size_t slot_stride)
{
    const bool norle = codec >= SPRINTZ_CODEC_DELTA_NORLE;
    return !norle && !col_stride && D <= 64 && lat_chunk_fits(true, esz, nchunks, chunk_len, D) &&
           ((uintptr_t)d_src % 16) == 0 && (nchunks == 1 || ((uint64_t)chunk_len * esz) % 16 == 0) && slot_stride % 16 == 0 && ((uintptr_t)d_slots % 16) == 0 &&
           // (a chunk is read in 16-byte pieces from a 16-byte aligned start: the last piece may reach past its end, never past the piece that holds its last byte)
           // (the encoder's crossover sits higher than the decoder's -- the lane-per-column encoders take ~100 us (uint16 x 8) / ~175 us (uint8 x 8)
           //  for ANY batch up to ~16 000 chunks: 75 vs 100 us at 3 072 chunks, 105 vs 101 at 4 096; 32 columns: 24 vs 26 at 1 024 -- tools/lat_sweep_enc.py)
           nchunks <= (uint64_t)process().lat_chunks.load(std::memory_order_relaxed) * (D > 16 ? 1u : 3u) / (D > 16 ? 3u : 2u) && !process().no_fast.load(std::memory_order_relaxed);
}

int encode_launch(int codec, int esz, const void* d_src, uint64_t total_len, uint32_t chunk_len, uint16_t ndims,
                  void* d_slots, size_t slot_stride, uint32_t* d_sizes, int64_t* d_rets, hipStream_t st, int write_size,
                  uint64_t col_stride = 0, int general = 0, DenseRequest* dense = nullptr, const HostCall* hc = nullptr)
{
    const uint64_t nchunks = sprintz_mi355x_num_chunks(total_len, chunk_len);
    if (nchunks == 0) return 0;
    const int D = ndims;
    const bool norle = codec >= SPRINTZ_CODEC_DELTA_NORLE;
    const bool lowdim = (norle || general) ? false : is_lowdim(esz, D);
    const Mapping m = choose_mapping(D, lowdim);
    const int DP = 1 << m.log2DP;

    EncodeArgs a{};
    a.src = d_src;
    a.total_len = total_len;
    a.chunk_len = chunk_len;
    a.nchunks = nchunks;
    a.D = D;
    a.log2DP = m.log2DP;
    a.slots = (uint8_t*)d_slots;
    a.slot_stride = slot_stride;
    a.sizes = d_sizes;
    a.rets = d_rets;
    a.write_size = write_size;
    a.col_stride = col_stride;
    a.norle = norle ? (codec == SPRINTZ_CODEC_XFF_NORLE ? 2 : 1) : 0;
    a.raw = codec == SPRINTZ_CODEC_BITPACK_NORLE ? 1 : 0;
    // 513 .. 2047 columns: one workgroup per chunk, the window holds one stream group (any_ndims.hip)
    if (D > 512) {
        if (norle || col_stride) return fail(SPRINTZ_E_UNSUPPORTED, "more than 512 columns: the RLE codecs, row-major only");
        if (nchunks > 0x7fffffffull) return fail(SPRINTZ_E_INVALID, "too many chunks for one launch");
        if (D > 2047) {                                        // column tiles, fields OR-ed straight into the zeroed slot (any_ndims.hip, "big")
            if (slot_stride % 16 || ((uintptr_t)d_slots & 15)) return fail(SPRINTZ_E_INVALID, "more than 2047 columns: slots must be 16-byte aligned and a multiple of 16 bytes");
            if (!is_plain_device_memory(d_slots)) return fail(SPRINTZ_E_INVALID, "more than 2047 columns: the slots must be device memory (hipMalloc), not mapped host or managed memory");
            int32_t* counters = nullptr;
            const bool fire = codec == SPRINTZ_CODEC_XFF;
            if (fire && (uint64_t)nchunks * (uint64_t)D * 4 > (1ull << 30)) return fail(SPRINTZ_E_UNSUPPORTED, "more than 2047 columns, FIRE: the counters' scratch (nchunks x ndims x 4 bytes) is limited to 1 GiB a launch: split the batch");
            if (fire && hipMallocAsync((void**)&counters, (size_t)nchunks * (size_t)D * 4, st) != hipSuccess) return fail(SPRINTZ_E_HIP, "hipMallocAsync of the counters' scratch (not available during stream capture)");
            const hipError_t eb = launch_encode_big(8 * esz, fire, (unsigned)nchunks, st, a, counters);
            if (counters) (void)hipFreeAsync(counters, st);
            if (eb != hipSuccess) return fail(SPRINTZ_E_HIP, "encode_big kernel launch", eb);
            return 0;
        }
        a.cap = ((uint32_t)group_bytes_max(esz, D) + 64u + 15u) & ~15u;
        const hipError_t ea = launch_encode_any(8 * esz, codec == SPRINTZ_CODEC_XFF, (unsigned)nchunks, a.cap, st, a);
        if (ea != hipSuccess) return fail(SPRINTZ_E_HIP, "encode_any kernel launch", ea);
        return 0;
    }
    // the generic kernel's window is a power-of-two RING flushed in 16-byte pieces; the kernels that flush whole 128-byte lines
    // (encode_fast.h, encode_wide.h) need that much more room in front of the write position: cap_drain (theirs alone -- added
    // to every encoder it doubled the generic ring wherever the group sat just under a power of two)
    a.cap = next_pow2((uint32_t)group_bytes_max(esz, D) + 48u);
    const uint32_t cap_drain = next_pow2((uint32_t)group_bytes_max(esz, D) + 48u + (uint32_t)(SPRINTZ_ENC_DRAIN_ALIGN - 16));
    const size_t shmem = ((size_t)a.cap + 16) * (kThreads / DP);
    if (shmem > 160 * 1024) return fail(SPRINTZ_E_UNSUPPORTED, "ndims too large for the LDS output ring");

    // Fast path (encode_fast.h): general layout, one column per lane, every 8 x D input
    // block 16-byte aligned, the power-of-two group at least half full.
    int fdp = 4;
    while (fdp < D) fdp <<= 1;
    const size_t blk_bytes = (size_t)8 * D * esz;
    const bool fast_common = !lowdim && !a.raw && D <= 64 && 2 * D > fdp && ((uintptr_t)d_src % 16) == 0 && !process().no_fast.load(std::memory_order_relaxed);
    const bool fast = col_stride ? fast_common && col_stride % 8 == 0 && (chunk_len / (uint32_t)D) % 8 == 0
                                 : fast_common && blk_bytes % 16 == 0 && ((uint64_t)chunk_len * esz) % 16 == 0;
    hipError_t e;
    if (hc && !encode_lat_fits(codec, esz, nchunks, chunk_len, D, col_stride, d_src, d_slots, slot_stride)) return fail(SPRINTZ_E_HIP, "internal: host call on a kernel that cannot end it");
    if (encode_lat_fits(codec, esz, nchunks, chunk_len, D, col_stride, d_src, d_slots, slot_stride)) {
        if (hc) { a.host_flag = hc->flag; a.host_ticket = hc->ticket; }
        int ldp = 4;
        while (ldp < D) ldp <<= 1;
        if (esz == 1 && ldp < 8 && !lowdim) ldp = 8;
        e = launch_encode_lat(8 * esz, codec == SPRINTZ_CODEC_XFF, ldp, lowdim, (unsigned)nchunks, (uint32_t)sprintz_mi355x_compress_bound(esz, chunk_len, ndims), st, a);
        if (e != hipSuccess) return fail(SPRINTZ_E_HIP, "encode_lat kernel launch", e);
        return 0;
    }
    // large batches of the DELTA codec, general layout, rows of whole 16-byte pieces: the block-parallel encoder (encode_blk.h) -- a thread
    // per (block, 16-byte row piece), the RLE state machine as scans; the container, if one was asked for, by the scan + copy passes.
    // (The container inside this launch -- images flushed straight to their place, found by compact_tail.h's chained scan -- was built and
    //  measured on BASELINE config 3 at 10 KB, 17 476 workgroups of three chunks: 0.296 ms against 0.267 for the launches in a row.  Taken
    //  apart: no scan, no tickets 0.177; tickets alone +0.070 (17 476 atomics on one word); the look-back alone +0.089 with 256 predecessors
    //  a hop, +0.114 with 1 024 -- a workgroup that lives 10 us waits for the slowest of a thousand resident predecessors with 33 KB of LDS
    //  held.  The tail pays from 64 chunks a workgroup on, as on the lane-per-column kernels.
    //  Second form: a workgroup takes 16 / 32 / 64 chunks in passes of three and ends with compact_tail.h's dense_tail (slot -> container copy, one
    //  chained-scan step per workgroup): 0.352 / 0.369 / 0.432 ms against 0.280 on the same box -- the looped kernel needs 149 registers (3 waves a
    //  SIMD instead of 4) and 820 - 3 277 workgroups are one to three cohorts: the tails do not hide behind anybody's encoding.
    //  Third form, priced before it was built: encoders that never wait -- they flush to their slots with write-through (sc0 sc1) stores, wait for
    //  them and add their size to their block's word; the block's last finisher finds the block's place (a look-back over a few hundred blocks) and
    //  copies it.  The encoder's side alone (the stores, the s_waitcnt, one atomic a chunk; no placement at all) measured 0.302 against 0.270 ms for
    //  the whole compress call: a third of the 0.098 ms the scan + copy launches cost is gone before the placers' copies and the last block's tail.)
    {
        const int blk_from = process().blk_chunks.load(std::memory_order_relaxed);
        const int blk_which = process().blk_kernels.load(std::memory_order_relaxed);
        if (blk_from > 0 && (blk_which & 1) && nchunks >= (uint64_t)blk_from && codec == SPRINTZ_CODEC_DELTA && !lowdim && !col_stride && !hc && write_size &&
            ((uintptr_t)d_src % 16) == 0 && slot_stride % 16 == 0 && ((uintptr_t)d_slots % 16) == 0 && !process().no_fast.load(std::memory_order_relaxed)) {
            const BlkEncGeom g = blk_enc_geom((uint32_t)esz, chunk_len, (uint32_t)D, (uint32_t)sprintz_mi355x_compress_bound(esz, chunk_len, ndims));
            if (g.ok) {
                const uint64_t bgrid = (nchunks + g.CPW - 1) / g.CPW;
                if (bgrid > 0x7fffffffull) return fail(SPRINTZ_E_INVALID, "too many chunks for one launch");
                e = launch_encode_blk(8 * esz, (unsigned)bgrid, st, a, g);
                if (e != hipSuccess) return fail(SPRINTZ_E_HIP, "encode_blk kernel launch", e);
                return 0;
            }
        }
        // the same for univariate streams of the low-dim layout (BASELINE config 1): a thread per 16 bytes of the series
        if (blk_from > 0 && (blk_which & 4) && nchunks >= (uint64_t)blk_from && codec == SPRINTZ_CODEC_DELTA && lowdim && D == 1 && !col_stride && !hc && write_size &&
            ((uintptr_t)d_src % 16) == 0 && slot_stride % 16 == 0 && ((uintptr_t)d_slots % 16) == 0 && !process().no_fast.load(std::memory_order_relaxed)) {
            const BlkEncGeom g = blk_enc_uni_geom((uint32_t)esz, chunk_len, (uint32_t)sprintz_mi355x_compress_bound(esz, chunk_len, ndims));
            if (g.ok) {
                const uint64_t bgrid = (nchunks + g.CPW - 1) / g.CPW;
                if (bgrid > 0x7fffffffull) return fail(SPRINTZ_E_INVALID, "too many chunks for one launch");
                e = launch_encode_blk_uni(8 * esz, (unsigned)bgrid, st, a, g);
                if (e != hipSuccess) return fail(SPRINTZ_E_HIP, "encode_blk_uni kernel launch", e);
                return 0;
            }
        }
    }
    // the container built inside the launch (compact_tail.h): every kernel of encode_fast.h / encode_wide.h carries the tail
    auto arm_dense = [&](uint64_t grid, size_t groups) -> int {
        // (column-major sources keep the two-launch path: with the tail in encode_fast<CM> BASELINE config 5 took 0.089 instead of 0.076 ms, its
        //  8 M-row form 0.407 instead of 0.359 -- eight chunks a workgroup make the chained scan eight times as long per byte as sixty-four do)
        //  Measured likewise on the row-major kernels: 8 uint16 columns (64 chunks a workgroup) 0.661 with the tail, 0.670 without; 16 columns (32 chunks)
        //  0.148 / 0.142; 64 columns (8) 0.158 / 0.144; BASELINE config 3 at 10 KB (8) 0.431 / 0.408 -- the tail pays from 64 chunks a workgroup on.
        if (!(dense && dense->d_dense && groups == 64) || col_stride) return 0;
        a.dn.dense = (uint8_t*)dense->d_dense;
        a.dn.offsets = dense->d_offsets;
        a.dn.wg_state = (uint64_t*)dense->d_tmp;
        a.dn.grid = (uint32_t)grid;
        HIP_TRY(hipMemsetAsync(dense->d_tmp, 0, ((size_t)grid + 1) * sizeof(uint64_t), st));   // look-back words + the ticket counter
        dense->fused = true;
        return 0;
    };
    // two columns per lane for narrow row-major streams too (encode_wide.h with 4 .. 32 lanes a chunk): fewer instructions per sample
    // than encode_fast.h's one column per lane on every shape measured (tools/enc_pair_sweep.sh: -6 % .. -35 %)
    // (not for a handful of chunks: there a chunk's latency is what counts, and half the lanes per chunk make it longer -- a single 10 KB
    //  sprintz_compress_xff_16b call 127 us against 111 with one column per lane; from a thousand chunks on the two are level or better)
    const int pair_from = process().enc_pair.load(std::memory_order_relaxed);
    // (column-major sources too: encode_fast.h's bursts, two columns' blocks per lane)
    const bool pair_layout = col_stride ? col_stride % 8 == 0 && (chunk_len / (uint32_t)D) % 8 == 0
                                        : blk_bytes % 16 == 0 && ((uint64_t)chunk_len * esz) % 16 == 0;
    if (pair_from > 0 && nchunks >= (uint64_t)pair_from && fast_common && D >= 5 && pair_layout && (uint64_t)chunk_len * esz >= 2 * blk_bytes) {
        int pdp = 4;
        while (2 * pdp < D) pdp <<= 1;
        const size_t pgroups = kThreads / pdp;
        // the window as long as it must be (the linear window needs no power of two): 592 instead of 672 bytes a chunk at 8 uint16 columns,
        // four workgroups a CU instead of three
        a.cap = ((uint32_t)group_bytes_max(esz, D) + 48u + (uint32_t)(SPRINTZ_ENC_DRAIN_ALIGN - 16) + 15u) & ~15u;
        // input staging: one 8 x D block (row-major: LDS transpose) or a burst of 4 blocks x (2 * pdp) columns (column-major)
        const size_t pstage = col_stride ? (size_t)4 * (2 * pdp) * (esz == 2 ? 16 : 8) : ((blk_bytes + 15) & ~(size_t)15);
        a.lds_group_stride = (uint32_t)(a.cap + pstage + 16);
        if ((a.lds_group_stride / 16) % 2 == 0) a.lds_group_stride += 16;    // an odd number of 16-byte units: the chunks of a wavefront start on different banks
        const uint64_t pgrid = (nchunks * (uint64_t)pdp + kThreads - 1) / kThreads;
        if (pgrid > 0x7fffffffull) return fail(SPRINTZ_E_INVALID, "too many chunks for one launch");
        if (int rc = arm_dense(pgrid, pgroups)) return rc;
        const bool pfire = codec == SPRINTZ_CODEC_XFF || codec == SPRINTZ_CODEC_XFF_NORLE;
        // (experiment knob: extra, unused LDS a workgroup claims -- fewer resident waves; what occupancy is worth to this loop: DESIGN 4.4)
        static const size_t lds_pad = getenv("SPRINTZ_MI355X_ENC_LDS_PAD") ? (size_t)atol(getenv("SPRINTZ_MI355X_ENC_LDS_PAD")) : 0;
        e = esz == 1 ? launch_encode_pair_w8(pfire, pdp, D == 2 * pdp, (unsigned)pgrid, (size_t)a.lds_group_stride * pgroups + lds_pad, st, a)
                     : launch_encode_pair_w16(pfire, pdp, D == 2 * pdp, (unsigned)pgrid, (size_t)a.lds_group_stride * pgroups + lds_pad, st, a);
        if (e != hipSuccess) return fail(SPRINTZ_E_HIP, "encode_wide (pair) kernel launch", e);
        return 0;
    }
    if (fast) {
        const size_t fgroups = kThreads / fdp;
        a.cap = cap_drain;
        // input staging: one 8 x D block (row-major: LDS transpose) or two bursts of 4 blocks x fdp columns (column-major)
        const size_t in_stage = col_stride ? (size_t)4 * fdp * (esz == 2 ? 16 : 8) : ((blk_bytes + 15) & ~(size_t)15);
        a.lds_group_stride = (uint32_t)(a.cap + in_stage + 16);
        const size_t fshmem = (size_t)a.lds_group_stride * fgroups;
        const uint64_t fthreads = nchunks * (uint64_t)fdp;
        const uint64_t fgrid = (fthreads + kThreads - 1) / kThreads;
        if (fgrid > 0x7fffffffull) return fail(SPRINTZ_E_INVALID, "too many chunks for one launch");
        if (int rc = arm_dense(fgrid, fgroups)) return rc;
        e = esz == 1 ? launch_encode_fast_w8((codec == SPRINTZ_CODEC_XFF || codec == SPRINTZ_CODEC_XFF_NORLE), fdp, D == fdp, (unsigned)fgrid, fshmem, st, a)
                     : launch_encode_fast_w16((codec == SPRINTZ_CODEC_XFF || codec == SPRINTZ_CODEC_XFF_NORLE), fdp, D == fdp, (unsigned)fgrid, fshmem, st, a);
        if (e != hipSuccess) return fail(SPRINTZ_E_HIP, "encode_fast kernel launch", e);
        return 0;
    }
    // streams of 65 .. 128 columns (BASELINE config 3): two columns per lane (encode_wide.h)
    if (!lowdim && !a.raw && !col_stride && D > 64 && D <= 128 && blk_bytes % 16 == 0 && ((uint64_t)chunk_len * esz) % 16 == 0 &&
        (uint64_t)chunk_len * esz >= 2 * blk_bytes && ((uintptr_t)d_src % 16) == 0 && !process().no_fast.load(std::memory_order_relaxed)) {
        // (8 bits, 65 .. 80 columns: 32 lanes a chunk -- a pair + a single column per lane -- two chunks a wavefront)
        const bool wsplit = esz == 1 && D <= 80 && process().split_lanes.load(std::memory_order_relaxed) != 0;
        const size_t wlanes = wsplit ? 32 : 64, wgroups = kThreads / wlanes;
        a.cap = cap_drain;
        a.lds_group_stride = (uint32_t)(a.cap + ((blk_bytes + 15) & ~(size_t)15) + 16);
        const uint64_t wgrid = (nchunks * (uint64_t)wlanes + kThreads - 1) / kThreads;
        if (wgrid > 0x7fffffffull) return fail(SPRINTZ_E_INVALID, "too many chunks for one launch");
        if (int rc = arm_dense(wgrid, wgroups)) return rc;
        const bool wfire = codec == SPRINTZ_CODEC_XFF || codec == SPRINTZ_CODEC_XFF_NORLE;
        e = wsplit   ? launch_encode_split_w8(wfire, (unsigned)wgrid, (size_t)a.lds_group_stride * wgroups, st, a)
          : esz == 1 ? launch_encode_wide_w8(wfire, D == 128, (unsigned)wgrid, (size_t)a.lds_group_stride * wgroups, st, a)
                     : launch_encode_wide_w16(wfire, D == 128, (unsigned)wgrid, (size_t)a.lds_group_stride * wgroups, st, a);
        if (e != hipSuccess) return fail(SPRINTZ_E_HIP, "encode_wide kernel launch", e);
        return 0;
    }
    // univariate streams: one lane per chunk, quad-loaded 64-byte input windows, 64-byte output units (encode_uni.h)
    // (and the other low-dim shapes: 2 columns, 3 and 4 at 8 bits)
    if (lowdim && (D <= 2 || esz == 1) && !col_stride && !process().no_fast.load(std::memory_order_relaxed)) {
        const uint64_t ugrid = (nchunks + 255) / 256;
        if (ugrid > 0x7fffffffull) return fail(SPRINTZ_E_INVALID, "too many chunks for one launch");
        // (round 6: the container inside THIS launch was built too -- 256 chunks a workgroup, 2 048 workgroups on BASELINE config 1, a short chain --
        //  and measured: 0.405 ms with a lane-parallel copy (a piece's chunk found by bisection), 0.49 - 0.51 chunk by chunk, against 0.3945 for
        //  encode + scan + copy in a row: streams of ~440 bytes re-read from their slots cost the workgroup what the copy pass costs.  Not kept.)
        e = esz == 1 ? launch_encode_uni_w8(codec == SPRINTZ_CODEC_XFF, D, (unsigned)ugrid, st, a)
                     : launch_encode_uni_w16(codec == SPRINTZ_CODEC_XFF, D, (unsigned)ugrid, st, a);
        if (e != hipSuccess) return fail(SPRINTZ_E_HIP, "encode_uni kernel launch", e);
        return 0;
    }
    const uint64_t threads = nchunks * (uint64_t)DP;
    const uint64_t grid = (threads + kThreads - 1) / kThreads;
    if (grid > 0x7fffffffull) return fail(SPRINTZ_E_INVALID, "too many chunks for one launch");
    e = esz == 1 ? launch_encode_w8((codec == SPRINTZ_CODEC_XFF || codec == SPRINTZ_CODEC_XFF_NORLE), lowdim, m.cpl, (unsigned)grid, shmem, st, a)
                 : launch_encode_w16((codec == SPRINTZ_CODEC_XFF || codec == SPRINTZ_CODEC_XFF_NORLE), lowdim, m.cpl, (unsigned)grid, shmem, st, a);
    if (e != hipSuccess) return fail(SPRINTZ_E_HIP, "encode kernel launch", e);
    return 0;
}

// RAII device buffer for the host-pointer wrappers
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 1); }
};

// Host framing walk: how many bytes does this stream span and how many
// elements does it decode to?  Needed because the reference's decompress()
// signature carries no length (sprintz.h:20) but an H2D copy must be sized.
// Touches headers and run lengths only -- no sample is decoded here.
//
// The walk trusts nothing it reads: it stops with `false` as soon as the framing claims more than
// one call may carry -- kMaxCallElems decoded elements or kMaxCallBytes of stream -- so a garbage
// group count or run length cannot send it (or the H2D copy sized from it) across the address
// space.  (The reference has no such limit: it reads wherever its header points.)
constexpr uint64_t kMaxCallElems = 1ull << 31;
constexpr uint64_t kMaxCallBytes = 1ull << 32;
bool walk_stream(const uint8_t* s, int esz, int D, uint32_t ngroups, uint32_t remaining, bool lowdim,
                 uint64_t* nbytes, uint64_t* nelems, bool norle = false)
{
    const int W = 8 * esz, HB = esz == 1 ? 3 : 4;
    const uint32_t hdr_bytes = (2u * D * HB + 7u) / 8u;
    uint64_t pos = 0, blocks = 0;
    auto field = [&](const uint8_t* h, uint32_t idx) {
        const uint32_t bit = idx * HB;
        uint32_t x = h[bit >> 3];
        if (((bit & 7) + HB) > 8) x |= (uint32_t)h[(bit >> 3) + 1] << 8;
        return (x >> (bit & 7)) & ((1u << HB) - 1);
    };
    for (uint32_t g = 0; g < ngroups; g++) {
        if (pos > kMaxCallBytes || blocks * 8ull * D > kMaxCallElems) return false;
        const uint8_t* h = s + pos;
        pos += hdr_bytes;
        for (int slot = 0; slot < 2; slot++) {
            uint32_t total = 0;
            for (int d = 0; d < D; d++) {
                uint32_t f = field(h, slot * D + d);
                total += (f == (uint32_t)(W - 1)) ? (uint32_t)W : f;
            }
            if (total == 0 && norle) {
                blocks += 1;                                   // a block of zeros has no payload
            } else if (total == 0) {
                uint32_t b0 = s[pos++], len = b0 & 0x7f;
                if (b0 & 0x80) len |= (uint32_t)s[pos++] << 7;
                blocks += len;
            } else {
                pos += lowdim ? total : 8ull * ((total + 7) / 8);
                blocks += 1;
            }
        }
    }
    *nbytes = pos + (uint64_t)remaining * esz;
    *nelems = blocks * 8ull * D + remaining;
    return *nbytes <= kMaxCallBytes && *nelems <= kMaxCallElems;
}

// ---- per-thread scratch of the host-pointer entry points -------------------------------------
// lzbench drives the single-call symbols once per 10 KB block, so they must not pay for
// hipMalloc/hipFree and pageable copies on every call: each thread keeps ONE device buffer and
// ONE pinned staging buffer that only grow, and a private non-blocking stream.  A call is then
// memcpy -> one H2D -> kernel -> one D2H -> memcpy, one stream synchronisation.  A thread that
// ends hands its scratch to a process-wide free list (no HIP call runs in a thread_local
// destructor; nothing is freed before the process ends).
struct Scratch {
    int device = -1;
    hipStream_t stream = nullptr;
    uint8_t* dev = nullptr;
    size_t dev_cap = 0;
    uint8_t* pin = nullptr;              // hipHostMallocMapped | Coherent: the kernels of a single call read and write it directly
    uint8_t* pin_dev = nullptr;          // the same bytes as the device sees them (hipHostGetDevicePointer)
    size_t pin_cap = 0;
    hipEvent_t done_spin = nullptr;      // waited for by spinning (a shared stream: the call waits for ITS launches, not for the stream)
    bool shared_stream = false;
    uint64_t* flag = nullptr;            // one mapped host word: the flag kernel that ends a polled call writes the call's ticket here
    uint64_t* flag_dev = nullptr;
    uint64_t ticket = 0;
    uint64_t last_wait_ns = 0;           // how long the last polled call of this thread waited
};
constexpr size_t kPinMax = 4u << 20;     // larger transfers go straight from/to the caller's memory

// ---- waiting for a single call's launches ---------------------------------------------------------
// hipStreamSynchronize spins: the shortest wait there is while the waiting threads have cores to spin on.  With more
// callers inside the library than that (lzbench -T64 in a 16-CPU container: 64 spinners on 16 CPUs' worth of quota get
// throttled, and the one whose kernel HAS finished waits for a time slice) a call ends with a kernel that writes the call's
// ticket into a mapped host word, and the caller sleeps and polls that word instead (a blocking-sync event was no better
// than spinning: the runtime's wait is where the CPU time went).  SPRINTZ_OPT_HOST_WAIT: 0 = by the number of callers (default), 1 = always spin, 2 = always sleep.
std::atomic<int> g_calls_inside{0};
int spin_budget()
{
    static const int n = [] {
        long q = (long)std::thread::hardware_concurrency();
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {           // cgroup v2 CPU quota, if any
            long quota = 0, period = 0;
            if (fscanf(f, "%ld %ld", &quota, &period) == 2 && quota > 0 && period > 0) q = std::min(q > 0 ? q : quota / period, quota / period);
            fclose(f);
        }
        return (int)std::min(4l, std::max(1l, q / 2));                   // half of them (the callers do other work too), and no more than 4: from 5
                                                                         // callers on the sleeping wait is the faster one (tools/mt_cmd.sh)
    }();
    return n;
}
struct CallGuard {
    int n;
    CallGuard() : n(g_calls_inside.fetch_add(1, std::memory_order_relaxed) + 1) {}
    ~CallGuard() { g_calls_inside.fetch_sub(1, std::memory_order_relaxed); }
};

std::mutex g_scratch_mu;
std::vector<Scratch*> g_scratch_free;
std::map<int, std::vector<hipStream_t>> g_stream_pool;      // per device: the streams the host-pointer calls share
std::map<int, size_t> g_stream_next;

struct ScratchHolder {
    Scratch* s = nullptr;
    ~ScratchHolder()
    {
        if (!s) return;
        std::lock_guard<std::mutex> lk(g_scratch_mu);
        g_scratch_free.push_back(s);
    }
};
thread_local ScratchHolder t_scratch;

size_t round_up(size_t x, size_t a) { return (x + a - 1) & ~(a - 1); }

// this thread's scratch on the current device, with at least dev_bytes / pin_bytes of room
int acquire_scratch(size_t dev_bytes, size_t pin_bytes, Scratch** out)
{
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    Scratch* sc = t_scratch.s;
    if (sc && sc->device != dev) {                       // the thread moved to another device
        std::lock_guard<std::mutex> lk(g_scratch_mu);
        g_scratch_free.push_back(sc);
        sc = t_scratch.s = nullptr;
    }
    if (!sc) {
        {
            std::lock_guard<std::mutex> lk(g_scratch_mu);
            for (size_t i = 0; i < g_scratch_free.size(); i++)
                if (g_scratch_free[i]->device == dev) {
                    sc = g_scratch_free[i];
                    g_scratch_free.erase(g_scratch_free.begin() + (long)i);
                    break;
                }
        }
        if (!sc) {
            sc = new Scratch();
            sc->device = dev;
            // The threads of a process share a FEW streams (SPRINTZ_OPT_HOST_STREAMS, default 4 = the hardware queues the runtime
            // maps streams onto): with a private stream per thread, 64 callers made 64 streams that the runtime multiplexes over
            // its 4 queues with a barrier packet at every switch -- 64 threads got a third of the calls per second of 8.
            const int k = process().host_streams.load(std::memory_order_relaxed);
            hipError_t e = hipSuccess;
            if (k > 0) {
                std::lock_guard<std::mutex> lk(g_scratch_mu);
                auto& pool = g_stream_pool[dev];
                if ((int)pool.size() < k) {
                    hipStream_t st = nullptr;
                    e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
                    if (e == hipSuccess) pool.push_back(st);
                }
                if (e == hipSuccess) { sc->stream = pool[g_stream_next[dev]++ % pool.size()]; sc->shared_stream = true; }
            } else {
                e = hipStreamCreateWithFlags(&sc->stream, hipStreamNonBlocking);
            }
            if (e != hipSuccess) { delete sc; return fail(SPRINTZ_E_HIP, "hipStreamCreateWithFlags", e); }
            const char* what = "hipEventCreateWithFlags";
            e = hipEventCreateWithFlags(&sc->done_spin, hipEventDisableTiming);
            if (e != hipSuccess) sc->done_spin = nullptr;
            if (e == hipSuccess) { what = "hipHostMalloc (completion flag)"; e = hipHostMalloc((void**)&sc->flag, 64, hipHostMallocMapped | hipHostMallocCoherent); if (e != hipSuccess) sc->flag = nullptr; }
            if (e == hipSuccess) { *sc->flag = 0; what = "hipHostGetDevicePointer (completion flag)"; e = hipHostGetDevicePointer((void**)&sc->flag_dev, sc->flag, 0); }
            if (e != hipSuccess) {
                if (sc->flag) (void)hipHostFree(sc->flag);
                if (sc->done_spin) (void)hipEventDestroy(sc->done_spin);
                if (!sc->shared_stream) (void)hipStreamDestroy(sc->stream);
                delete sc;
                return fail(SPRINTZ_E_HIP, what, e);
            }
        }
        t_scratch.s = sc;
    }
    if (sc->dev_cap < dev_bytes) {
        HIP_TRY(hipStreamSynchronize(sc->stream));
        if (sc->dev) (void)hipFree(sc->dev);
        sc->dev = nullptr; sc->dev_cap = 0;
        const size_t want = round_up(dev_bytes + dev_bytes / 2, 1u << 16);
        HIP_TRY(hipMalloc((void**)&sc->dev, want));
        sc->dev_cap = want;
    }
    if (sc->pin_cap < pin_bytes) {
        HIP_TRY(hipStreamSynchronize(sc->stream));
        if (sc->pin) (void)hipHostFree(sc->pin);
        sc->pin = nullptr; sc->pin_cap = 0;
        const size_t want = round_up(pin_bytes + pin_bytes / 2, 1u << 16);
        HIP_TRY(hipHostMalloc((void**)&sc->pin, want, hipHostMallocMapped | hipHostMallocCoherent));
        sc->pin_cap = want;
        HIP_TRY(hipHostGetDevicePointer((void**)&sc->pin_dev, sc->pin, 0));
    }
    *out = sc;
    return 0;
}

// ends a polled call: one word, the call's ticket, into the thread's mapped flag (after the codec kernel in stream order; a kernel's
// end makes its stores to host memory visible before the next kernel of the stream starts)
__global__ void flag_kernel(uint64_t* flag, uint64_t ticket)
{
    __hip_atomic_store(flag, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// sleep, then look at the thread's flag word until it holds `want`: no runtime wait at all (tools/mt_cmd.sh, 16-CPU container:
// 64 threads 200k calls/s this way, 48k on a blocking-sync event, 52k spinning).  spin: look without sleeping (few callers).
int wait_flag(Scratch* sc, uint64_t want, bool spin)
{
    const auto t0 = std::chrono::steady_clock::now();
    if (spin) {
        for (uint32_t it = 1;; ++it) {
            if (__atomic_load_n(sc->flag, __ATOMIC_ACQUIRE) == want) break;
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#endif
            if ((it & 0xfffu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) {
                HIP_TRY(hipStreamSynchronize(sc->stream));               // a fault surfaces here
                if (__atomic_load_n(sc->flag, __ATOMIC_ACQUIRE) != want) return fail(SPRINTZ_E_HIP, "the call's completion flag was never written");
                break;
            }
        }
    } else {
        // the default 50 us of timer slack would make every 20 us sleep a 75 us one: 1 us for the duration of THIS wait, the caller's
        // own value put back before returning (it is the application's thread, not ours)
        struct SlackGuard {
            long old;
            SlackGuard() : old(prctl(PR_GET_TIMERSLACK, 0, 0, 0, 0)) { if (old > 1000) (void)prctl(PR_SET_TIMERSLACK, 1000ul, 0, 0, 0); }
            ~SlackGuard() { if (old > 1000) (void)prctl(PR_SET_TIMERSLACK, (unsigned long)old, 0, 0, 0); }
        } slack_guard;
        // first sleep: three quarters of what the thread's last call waited (with many callers a call queues behind the others' for
        // hundreds of microseconds), never under 15 us (no call is shorter); then 5, 10, 20 ... 160 us: 128 threads that each woke
        // every 5 us would spend the container's CPUs on waking up
        struct timespec ts{0, (long)std::min<uint64_t>(std::max<uint64_t>(15000, sc->last_wait_ns * 3 / 4), 2000000)};
        long step = 5000;
        for (uint32_t it = 0;; ++it) {
            nanosleep(&ts, nullptr);
            if (__atomic_load_n(sc->flag, __ATOMIC_ACQUIRE) == want) break;
            ts.tv_nsec = step;
            step = std::min(step * 2, 160000l);
            if ((it & 15) == 15 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) {
                HIP_TRY(hipStreamSynchronize(sc->stream));
                if (__atomic_load_n(sc->flag, __ATOMIC_ACQUIRE) != want) return fail(SPRINTZ_E_HIP, "the call's completion flag was never written");
                break;
            }
        }
        sc->last_wait_ns = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    }
    if ((want & 63) == 0) (void)hipStreamQuery(sc->stream);          // the runtime retires finished launches when asked about them
    return 0;
}

bool spin_wait(int callers)
{
    const int mode = process().host_wait.load(std::memory_order_relaxed);
    return mode == 1 || (mode == 0 && callers <= spin_budget());
}

// wait for everything this call put on the thread's stream
int wait_call(Scratch* sc, int callers)
{
    if (spin_wait(callers)) {
        if (sc->shared_stream) {
            HIP_TRY(hipEventRecord(sc->done_spin, sc->stream));
            HIP_TRY(hipEventSynchronize(sc->done_spin));
        } else {
            HIP_TRY(hipStreamSynchronize(sc->stream));
        }
        return 0;
    }
    const uint64_t want = ++sc->ticket;
    hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(1), 0, sc->stream, sc->flag_dev, want);
    HIP_TRY(hipGetLastError());
    return wait_flag(sc, want, false);
}

// A single call's input, host -> HBM, as ONE wide read of the thread's mapped staging buffer (every lane a 16-byte piece,
// all requests of the call in flight over PCIe at once): the codec kernels walk their streams in dependent steps, which
// must find them in HBM/L2 -- a PCIe round trip per step would cost more than the whole call.  Launched on the call's
// stream right before the codec kernel; n16 = 16-byte pieces.
typedef uint32_t stage_v4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) stage_in_kernel(const stage_v4* __restrict__ host, stage_v4* __restrict__ dev, uint32_t n16)
{
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n16; i += gridDim.x * 256u)
        dev[i] = __builtin_nontemporal_load(host + i);
}
int stage_in(Scratch* sc, size_t pin_off, size_t dev_off, size_t bytes)
{
    const uint32_t n16 = (uint32_t)((bytes + 15) / 16);
    const unsigned grid = std::min<unsigned>((n16 + 255u) / 256u, 1024u);
    hipLaunchKernelGGL(stage_in_kernel, dim3(grid), dim3(256), 0, sc->stream, (const stage_v4*)(sc->pin_dev + pin_off), (stage_v4*)(sc->dev + dev_off), n16);
    HIP_TRY(hipGetLastError());
    return 0;
}

int64_t compress_host(int codec, int esz, const void* src, uint32_t len, void* dest, uint16_t ndims, int write_size,
                      int layout = SPRINTZ_LAYOUT_AUTO)
{
    if (ndims == 0) { fail(SPRINTZ_E_INVALID, "ndims == 0"); return -1; }          // sprintz.cpp:36
    int rc = check_common(codec, esz, ndims);
    if (rc) return rc;
    if (layout < SPRINTZ_LAYOUT_AUTO || layout > SPRINTZ_LAYOUT_LOWDIM || (layout && codec > SPRINTZ_CODEC_XFF))
        return fail(SPRINTZ_E_INVALID, "layout must be 0 (by ndims), 1 (general) or 2 (low-dim), RLE codecs only");
    if (layout == SPRINTZ_LAYOUT_LOWDIM && !is_lowdim(esz, ndims)) {               // sprintz_delta_lowdim.cpp:64-70
        fail(SPRINTZ_E_INVALID, "the low-dim layout takes ndims <= 4 at 8 bits, <= 2 at 16 bits");
        return -1;
    }
    if (len > (1u << 30)) return fail(SPRINTZ_E_UNSUPPORTED, "single call limited to 2^30 elements");
    if ((rc = ensure_device())) return rc;
    if (len == 0 && codec == SPRINTZ_CODEC_XFF_NORLE) {     // u64 0 with ndims in bytes 6..7 (sprintz_xff.cpp:58-63)
        uint8_t h[8] = {0, 0, 0, 0, 0, 0, (uint8_t)(ndims & 0xff), (uint8_t)(ndims >> 8)};
        memcpy(dest, h, 8);
        return 8;
    }
    if (len == 0 && codec >= SPRINTZ_CODEC_DELTA_NORLE) {   // {u32 0; u16 ndims} (format.h:65-72)
        uint8_t h[6] = {0, 0, 0, 0, (uint8_t)(ndims & 0xff), (uint8_t)(ndims >> 8)};
        memcpy(dest, h, 6);
        return 6 / esz;
    }
    if (len == 0) {   // zero elements: header only (reference: :116-124 with len == 0)
        uint8_t h[8] = {0};
        h[6] = (uint8_t)(ndims & 0xff); h[7] = (uint8_t)(ndims >> 8);
        if (write_size) memcpy(dest, h, 8);
        return write_size ? 8 / esz : 0;
    }
    const size_t bound = sprintz_mi355x_compress_bound(esz, len, ndims);
    const size_t src_bytes = (size_t)len * esz;
    CallGuard inside;
    // ---- the zero-copy call (everything that fits the staging buffer; the RLE codecs, whose encoders only WRITE their slot):
    // memcpy into the mapped staging buffer -> stage_in + encoder on the thread's stream, the encoder's slot, size and return
    // value landing straight in the staging buffer -> ONE wait -> memcpy out.  No copy engine, no memset, no second round trip.
    // staging: [source | size, ret (16 B) | slot]      device: [source + read slack]
    if (codec <= SPRINTZ_CODEC_XFF && ndims <= 2047 && src_bytes + 16 + bound + 512 <= kPinMax) {      // (above 2047 columns the encoder ORs into its slot with device atomics: a slot in HBM)
        const size_t p_meta = round_up(src_bytes + 16, 256), p_slot = p_meta + 16;
        Scratch* sc = nullptr;
        if ((rc = acquire_scratch(round_up(src_bytes, 16) + SPRINTZ_MI355X_READ_SLACK, p_slot + bound, &sc))) return rc;
        memcpy(sc->pin, src, src_bytes);
        uint32_t size = 0xffffffffu;         // a kernel that never reports must read as an error, not as the last call's answer
        int64_t ret = -1;
        memcpy(sc->pin + p_meta, &size, 4);
        memcpy(sc->pin + p_meta + 8, &ret, 8);
        // one chunk the workgroup-per-chunk encoder takes: it reads the staging buffer itself (one wide read, as stage_in's) and
        // ends the call by writing the ticket -- ONE launch, no runtime wait
        if (encode_lat_fits(codec, esz, 1, len, ndims, 0, sc->pin_dev, sc->pin_dev + p_slot, bound)) {
            HostCall hc;
            hc.flag = sc->flag_dev;
            hc.ticket = ++sc->ticket;
            rc = encode_launch(codec, esz, sc->pin_dev, len, len, ndims, sc->pin_dev + p_slot, bound, (uint32_t*)(sc->pin_dev + p_meta),
                               (int64_t*)(sc->pin_dev + p_meta + 8), sc->stream, write_size, 0, layout == SPRINTZ_LAYOUT_GENERAL, nullptr, &hc);
            if (rc) return rc;
            if ((rc = wait_flag(sc, hc.ticket, spin_wait(inside.n)))) { (void)hipStreamSynchronize(sc->stream); *sc->flag = 0; return rc; }   // nothing of this call may still target sc->pin / sc->flag when the scratch is reused
        } else {
            if ((rc = stage_in(sc, 0, 0, src_bytes))) return rc;
            rc = encode_launch(codec, esz, sc->dev, len, len, ndims, sc->pin_dev + p_slot, bound, (uint32_t*)(sc->pin_dev + p_meta),
                               (int64_t*)(sc->pin_dev + p_meta + 8), sc->stream, write_size, 0, layout == SPRINTZ_LAYOUT_GENERAL);
            if (rc) { (void)hipStreamSynchronize(sc->stream); return rc; }   // stage_in may still be reading sc->pin
            if ((rc = wait_call(sc, inside.n))) return rc;
        }
        memcpy(&size, sc->pin + p_meta, 4);
        memcpy(&ret, sc->pin + p_meta + 8, 8);
        if (size > bound) return fail(SPRINTZ_E_HIP, "encoder reported a size above its bound");
        memcpy(dest, sc->pin + p_slot, size);
        return ret;
    }
    // ---- larger calls and the run-less codecs -- device scratch: [source + read slack | size, ret (16 B) | slot]
    const size_t o_meta = round_up(src_bytes + SPRINTZ_MI355X_READ_SLACK, 256), o_slot = o_meta + 16;
    const bool pin_in = src_bytes <= kPinMax, pin_out = 16 + bound <= kPinMax;
    Scratch* sc = nullptr;
    if ((rc = acquire_scratch(o_slot + bound, std::max(pin_in ? src_bytes : 0, pin_out ? 16 + bound : 0), &sc))) return rc;
    if (pin_in) {
        memcpy(sc->pin, src, src_bytes);
        HIP_TRY(hipMemcpyAsync(sc->dev, sc->pin, src_bytes, hipMemcpyHostToDevice, sc->stream));
    } else {
        HIP_TRY(hipMemcpyAsync(sc->dev, src, src_bytes, hipMemcpyHostToDevice, sc->stream));
    }
    uint32_t* d_size = (uint32_t*)(sc->dev + o_meta);
    int64_t* d_ret = (int64_t*)(sc->dev + o_meta + 8);
    // the scratch is reused: a kernel that never reports must read as an error (size 0xffffffff > bound), not as the last call's answer
    HIP_TRY(hipMemsetAsync(d_size, 0xff, 16, sc->stream));
    rc = encode_launch(codec, esz, sc->dev, len, len, ndims, sc->dev + o_slot, bound, d_size, d_ret, sc->stream, write_size, 0,
                       layout == SPRINTZ_LAYOUT_GENERAL);
    if (rc) { (void)hipStreamSynchronize(sc->stream); return rc; }   // the H2D copy out of sc->pin may still be in flight
    uint32_t size = 0;
    int64_t ret = 0;
    // small slots: size, ret and the whole slot in one copy (one round trip); large ones: the 16 bytes first, then exactly `size`
    // bytes -- a compressible multi-megabyte call would otherwise move 2-4x the stream over PCIe
    constexpr size_t kOneCopyMax = 64u << 10;
    if (pin_out && 16 + bound <= kOneCopyMax) {
        HIP_TRY(hipMemcpyAsync(sc->pin, sc->dev + o_meta, 16 + bound, hipMemcpyDeviceToHost, sc->stream));
        HIP_TRY(hipStreamSynchronize(sc->stream));
        memcpy(&size, sc->pin, 4);
        memcpy(&ret, sc->pin + 8, 8);
        if (size > bound) return fail(SPRINTZ_E_HIP, "encoder reported a size above its bound");
        memcpy(dest, sc->pin + 16, size);
    } else if (pin_out) {
        HIP_TRY(hipMemcpyAsync(sc->pin, sc->dev + o_meta, 16, hipMemcpyDeviceToHost, sc->stream));
        HIP_TRY(hipStreamSynchronize(sc->stream));
        memcpy(&size, sc->pin, 4);
        memcpy(&ret, sc->pin + 8, 8);
        if (size > bound) return fail(SPRINTZ_E_HIP, "encoder reported a size above its bound");
        HIP_TRY(hipMemcpyAsync(sc->pin, sc->dev + o_slot, size, hipMemcpyDeviceToHost, sc->stream));
        HIP_TRY(hipStreamSynchronize(sc->stream));
        memcpy(dest, sc->pin, size);
    } else {
        uint8_t meta[16];
        HIP_TRY(hipMemcpyAsync(meta, sc->dev + o_meta, 16, hipMemcpyDeviceToHost, sc->stream));
        HIP_TRY(hipStreamSynchronize(sc->stream));
        memcpy(&size, meta, 4);
        memcpy(&ret, meta + 8, 8);
        if (size > bound) return fail(SPRINTZ_E_HIP, "encoder reported a size above its bound");
        HIP_TRY(hipMemcpyAsync(dest, sc->dev + o_slot, size, hipMemcpyDeviceToHost, sc->stream));
        HIP_TRY(hipStreamSynchronize(sc->stream));
    }
    return ret;
}

// shared tail of the single-call decoders: stream [s, s + nbytes) (host) -> nelems elements.
// device scratch: [offsets[2] (16 B) | stream + read slack | ret (16 B) | out]
int64_t decode_host_common(int codec, int esz, const uint8_t* s, uint64_t nbytes, uint64_t nelems, uint16_t ndims, void* dest,
                           int noheader, uint32_t ngroups, uint32_t remaining, const QuerySpec* qspec, uint64_t* result)
{
    const size_t o_out_meta = round_up(16 + nbytes + SPRINTZ_MI355X_READ_SLACK, 256), o_out = o_out_meta + 16;
    const size_t out_bytes = (size_t)nelems * esz;
    const bool want_out = !qspec || qspec->q != kQueryReduceOnly;
    CallGuard inside;
    // ---- the zero-copy call (plain decodes that fit the staging buffer): memcpy the stream into the mapped staging buffer ->
    // stage_in + decoder on the thread's stream, the decoder writing samples and its return value straight into the staging
    // buffer -> ONE wait -> memcpy out.   staging: [offsets[2] | stream | ret (16 B) | out]      device: [offsets[2] | stream + slack]
    if (!qspec && ndims <= 2047 && 16 + nbytes + 512 + 16 + out_bytes <= kPinMax) {      // (above 2047 columns the decoder reads its own output back: an output in HBM)
        const size_t p_ret = round_up(16 + nbytes + 16, 256), p_out = p_ret + 16;
        Scratch* sc = nullptr;
        int rc = acquire_scratch(round_up(16 + nbytes, 16) + SPRINTZ_MI355X_READ_SLACK, p_out + out_bytes, &sc);
        if (rc) return rc;
        const uint64_t meta[2] = {16, 16 + nbytes};
        int64_t ret = -1;
        memcpy(sc->pin, meta, 16);
        memcpy(sc->pin + 16, s, nbytes);
        memcpy(sc->pin + p_ret, &ret, 8);
        if (decode_lat_fits(codec, esz, 1, (uint32_t)nelems, ndims, noheader, QuerySpec{}, sc->pin_dev + p_out)) {   // (see compress_host)
            HostCall hc;
            hc.off0 = 16; hc.off1 = 16 + nbytes;
            hc.flag = sc->flag_dev;
            hc.ticket = ++sc->ticket;
            QuerySpec qs;
            qs.hc = &hc;
            rc = decode_launch(codec, esz, sc->pin_dev, nullptr, 1, (uint32_t)nelems, ndims, sc->pin_dev + p_out,
                               (int64_t*)(sc->pin_dev + p_ret), sc->stream, noheader, ngroups, remaining, qs);
            if (rc) return rc;
            if ((rc = wait_flag(sc, hc.ticket, spin_wait(inside.n)))) { (void)hipStreamSynchronize(sc->stream); *sc->flag = 0; return rc; }   // nothing of this call may still target sc->pin / sc->flag when the scratch is reused
        } else {
            if ((rc = stage_in(sc, 0, 0, 16 + nbytes))) return rc;
            rc = decode_launch(codec, esz, sc->dev, (const uint64_t*)sc->dev, 1, (uint32_t)nelems, ndims, sc->pin_dev + p_out,
                               (int64_t*)(sc->pin_dev + p_ret), sc->stream, noheader, ngroups, remaining, QuerySpec{});
            if (rc) { (void)hipStreamSynchronize(sc->stream); return rc; }
            if ((rc = wait_call(sc, inside.n))) return rc;
        }
        memcpy(&ret, sc->pin + p_ret, 8);
        if (ret < 0) return fail((int)ret, "decoder rejected the stream");
        if ((uint64_t)ret > nelems) return fail(SPRINTZ_E_CORRUPT, "decoder rejected the stream");
        memcpy(dest, sc->pin + p_out, (size_t)ret * esz);
        return ret;
    }
    const size_t o_res = round_up(o_out + (want_out ? out_bytes : 0), 256);
    const size_t res_bytes = qspec && qspec->qop ? (size_t)ndims * 8 : 0;
    const bool pin_in = 16 + nbytes <= kPinMax, pin_out = 16 + out_bytes <= kPinMax;
    Scratch* sc = nullptr;
    int rc = acquire_scratch(o_res + res_bytes, std::max<size_t>(std::max<size_t>(pin_in ? 16 + nbytes : 16, pin_out ? 16 + out_bytes : 16), res_bytes), &sc);
    if (rc) return rc;
    const uint64_t meta[2] = {16, 16 + nbytes};              // offsets[0], offsets[1] (= stream end) relative to the scratch
    if (pin_in) {
        memcpy(sc->pin, meta, 16);
        memcpy(sc->pin + 16, s, nbytes);
        HIP_TRY(hipMemcpyAsync(sc->dev, sc->pin, 16 + nbytes, hipMemcpyHostToDevice, sc->stream));
    } else {
        memcpy(sc->pin, meta, 16);
        HIP_TRY(hipMemcpyAsync(sc->dev, sc->pin, 16, hipMemcpyHostToDevice, sc->stream));
        HIP_TRY(hipMemcpyAsync(sc->dev + 16, s, nbytes, hipMemcpyHostToDevice, sc->stream));
    }
    int64_t* d_ret = (int64_t*)(sc->dev + o_out_meta);
    HIP_TRY(hipMemsetAsync(d_ret, 0xff, 8, sc->stream));      // a kernel that never reports reads as an error
    QuerySpec qs = qspec ? *qspec : QuerySpec{};
    if (res_bytes) {
        qs.qres = (uint64_t*)(sc->dev + o_res);
        HIP_TRY(hipMemsetAsync(qs.qres, 0, res_bytes, sc->stream));
    }
    rc = decode_launch(codec, esz, sc->dev, (const uint64_t*)sc->dev, 1, (uint32_t)nelems, ndims, want_out ? sc->dev + o_out : nullptr,
                       d_ret, sc->stream, noheader, ngroups, remaining, qs);
    if (rc) { (void)hipStreamSynchronize(sc->stream); return rc; }   // the H2D copy out of sc->pin may still be in flight
    int64_t ret = 0;
    if (want_out && pin_out) {                                // ret and the samples in one copy
        HIP_TRY(hipMemcpyAsync(sc->pin, sc->dev + o_out_meta, 16 + out_bytes, hipMemcpyDeviceToHost, sc->stream));
        HIP_TRY(hipStreamSynchronize(sc->stream));
        memcpy(&ret, sc->pin, 8);
        if (ret < 0) return fail((int)ret, "decoder rejected the stream");
        if ((uint64_t)ret > nelems) return fail(SPRINTZ_E_CORRUPT, "decoder rejected the stream");
        memcpy(dest, sc->pin + 16, (size_t)ret * esz);
    } else {
        HIP_TRY(hipMemcpyAsync(sc->pin, d_ret, 8, hipMemcpyDeviceToHost, sc->stream));
        HIP_TRY(hipStreamSynchronize(sc->stream));
        memcpy(&ret, sc->pin, 8);
        if (ret < 0) return fail((int)ret, "decoder rejected the stream");
        if ((uint64_t)ret > nelems) return fail(SPRINTZ_E_CORRUPT, "decoder rejected the stream");
        if (want_out) {
            HIP_TRY(hipMemcpyAsync(dest, sc->dev + o_out, (size_t)ret * esz, hipMemcpyDeviceToHost, sc->stream));
            HIP_TRY(hipStreamSynchronize(sc->stream));
        }
    }
    if (result && res_bytes) {
        HIP_TRY(hipMemcpyAsync(sc->pin, sc->dev + o_res, res_bytes, hipMemcpyDeviceToHost, sc->stream));
        HIP_TRY(hipStreamSynchronize(sc->stream));
        memcpy(result, sc->pin, res_bytes);
    }
    return ret;
}

int64_t decompress_host(int codec, int esz, const void* src, void* dest, int noheader, uint16_t nh_ndims,
                        uint32_t nh_ngroups, uint16_t nh_remaining, int layout = SPRINTZ_LAYOUT_AUTO)
{
    if (layout < SPRINTZ_LAYOUT_AUTO || layout > SPRINTZ_LAYOUT_LOWDIM || (layout && codec > SPRINTZ_CODEC_XFF))
        return fail(SPRINTZ_E_INVALID, "layout must be 0 (by ndims), 1 (general) or 2 (low-dim), RLE codecs only");
    const uint8_t* s = (const uint8_t*)src;
    uint32_t ngroups, remaining;
    uint16_t ndims;
    const bool norle = codec >= SPRINTZ_CODEC_DELTA_NORLE;
    if (norle) {                                               // {u32 len; u16 ndims}; sprintz_delta.cpp:803-807, :832
        uint32_t len;
        memcpy(&len, s, 4);
        memcpy(&ndims, s + (codec == SPRINTZ_CODEC_XFF_NORLE ? 6 : 4), 2);
        ngroups = (len < 128 || ndims == 0) ? 0 : len / (16u * ndims);
        remaining = len - ngroups * 16u * ndims;
        if (ndims == 0 && len == 0) return 0;
    } else if (!noheader) {
        uint16_t r16;
        memcpy(&ngroups, s, 4);
        memcpy(&r16, s + 4, 2);
        memcpy(&ndims, s + 6, 2);
        remaining = r16;
    } else {
        ngroups = nh_ngroups; remaining = nh_remaining; ndims = nh_ndims;
    }
    if (ndims == 0) { fail(SPRINTZ_E_INVALID, "ndims == 0"); return -1; }          // sprintz.cpp:36
    int rc = check_common(codec, esz, ndims);
    if (rc) return rc;
    if ((rc = ensure_device())) return rc;
    if (layout == SPRINTZ_LAYOUT_LOWDIM && !is_lowdim(esz, ndims)) {               // sprintz_delta_lowdim.cpp:423-429
        fail(SPRINTZ_E_INVALID, "the low-dim layout takes ndims <= 4 at 8 bits, <= 2 at 16 bits");
        return -1;
    }
    const bool general = layout == SPRINTZ_LAYOUT_GENERAL;
    const uint32_t hlen = norle ? (codec == SPRINTZ_CODEC_XFF_NORLE ? 8 : 6) : (noheader ? 0 : 8);
    uint64_t nbytes = 0, nelems = 0;
    if (!walk_stream(s + hlen, esz, ndims, ngroups, remaining, (norle || general) ? false : is_lowdim(esz, ndims), &nbytes, &nelems, norle))
        return fail(SPRINTZ_E_UNSUPPORTED, "stream framing exceeds the single-call limits (2^31 elements / 4 GiB): damaged header?");
    nbytes += hlen;
    if (nelems == 0) return 0;
    QuerySpec qs;
    qs.general = general ? 1 : 0;
    return decode_host_common(codec, esz, s, nbytes, nelems, ndims, dest, noheader, ngroups, remaining, general ? &qs : nullptr, nullptr);
}

// per-column reduction of the per-chunk partials.  A workgroup covers RB rows x D columns per
// pass (thread = (row r, column col): a row of partials is contiguous, so consecutive threads
// read consecutive words), folds its rows through LDS and issues ONE atomic per column --
// 64 K threads each doing their own atomic on 8 addresses took 166 us for 8 MB.
__global__ void __launch_bounds__(256) query_reduce_kernel(const uint64_t* partials, uint64_t nchunks, uint32_t D, int op,
                                                           unsigned long long* result)
{
    __shared__ unsigned long long acc[256];
    const uint32_t tid = threadIdx.x;
    const uint32_t RB = D >= 256 ? 1u : 256u / D;
    const uint32_t r = D >= 256 ? 0u : tid / D;
    const bool active = r < RB;
    for (uint32_t col = D >= 256 ? tid : tid % D; col < D; col += 256) {
        unsigned long long v = 0;
        if (active) {
            for (uint64_t c = (uint64_t)blockIdx.x * RB + r; c < nchunks; c += (uint64_t)gridDim.x * RB) {
                const unsigned long long x = partials[c * D + col];
                v = op == 1 ? (x > v ? x : v) : v + x;
            }
        }
        if (RB > 1) {
            acc[tid] = v;
            __syncthreads();
            if (r == 0) {
                for (uint32_t q = 1; q < RB; q++) {
                    const unsigned long long x = acc[q * D + col];
                    v = op == 1 ? (x > v ? x : v) : v + x;
                }
            }
            __syncthreads();
        }
        if (r == 0) {
            if (op == 1) atomicMax(&result[col], v);
            else atomicAdd(&result[col], v);
        }
    }
}

// single-call query (mirrors query_rowmajor_{delta,xff}_rle_{8b,16b}: sprintz_delta.h:95-98,
// sprintz_xff.h:90-93): host stream in, optional materialised data and per-column result out
int64_t query_host(int codec, int esz, const void* src, void* dest, int op, int materialize, uint64_t* result, int general)
{
    if (op < 0 || op > 2) return fail(SPRINTZ_E_INVALID, "op must be 0 (none), 1 (max) or 2 (sum)");
    if (materialize && !dest) return fail(SPRINTZ_E_INVALID, "materialize without a destination");
    const uint8_t* s = (const uint8_t*)src;
    uint32_t ngroups;
    uint16_t r16, ndims;
    memcpy(&ngroups, s, 4);
    memcpy(&r16, s + 4, 2);
    memcpy(&ndims, s + 6, 2);
    const uint32_t remaining = r16;
    if (ndims == 0) { fail(SPRINTZ_E_INVALID, "ndims == 0"); return -1; }
    int rc = check_common(codec, esz, ndims);
    if (rc) return rc;
    if ((rc = ensure_device())) return rc;
    uint64_t nbytes = 0, nelems = 0;
    if (!walk_stream(s + 8, esz, ndims, ngroups, remaining, general ? false : is_lowdim(esz, ndims), &nbytes, &nelems))
        return fail(SPRINTZ_E_UNSUPPORTED, "stream framing exceeds the single-call limits (2^31 elements / 4 GiB): damaged header?");
    nbytes += 8;
    if (result) memset(result, 0, (size_t)ndims * 8);
    if (nelems == 0) return 0;
    QuerySpec qs;
    qs.q = materialize ? (op ? kQueryMaterialize : kQueryOff) : kQueryReduceOnly;
    qs.qop = op;
    qs.general = general;
    return decode_host_common(codec, esz, s, nbytes, nelems, ndims, dest, 0, 0, 0, &qs, result);
}

}  // namespace

namespace sprintz {
// what the other translation units' host-pointer entry points (online.hip) share with this one: the process-wide
// device probe and the calling thread's pooled scratch (one device buffer, one pinned buffer, one non-blocking stream)
bool have_device() { return process().have_device; }
int host_scratch(size_t dev_bytes, size_t pin_bytes, HostScratch* out)
{
    Scratch* sc = nullptr;
    const int rc = acquire_scratch(dev_bytes, pin_bytes, &sc);
    if (rc) return rc;
    out->stream = sc->stream;
    out->dev = sc->dev;
    out->pin = sc->pin;
    return 0;
}
}  // namespace sprintz

// =============================================================== exported C-ABI
extern "C" {

int sprintz_mi355x_abi_version(void) { return SPRINTZ_MI355X_ABI_VERSION; }

int sprintz_mi355x_set_option(int option, int value)
{
    if (option == SPRINTZ_OPT_NO_FAST) { process().no_fast = value ? 1 : 0; return 0; }
    if (option == SPRINTZ_OPT_DENSE_MODE) {
        if (value < 0 || value > 1) return fail(SPRINTZ_E_INVALID, "dense mode must be 0 or 1");
        process().dense_mode = value;
        return 0;
    }
    if (option == SPRINTZ_OPT_HUF0_BIG_BATCH) {
        if (value < 0) return fail(SPRINTZ_E_INVALID, "the batch size must not be negative");
        sprintz::huf0_big_batch() = value;
        return 0;
    }
    if (option == SPRINTZ_OPT_HUF0_SYNC_CHUNKS) {
        if (value < 0) return fail(SPRINTZ_E_INVALID, "the batch size must not be negative");
        sprintz::huf0_sync_chunks() = value;
        return 0;
    }
    if (option == SPRINTZ_OPT_SPLIT_LANES) { process().split_lanes = value ? 1 : 0; return 0; }
    if (option == SPRINTZ_OPT_ENC_PAIR) {
        if (value < 0) return fail(SPRINTZ_E_INVALID, "the batch size must not be negative");
        process().enc_pair = value;
        return 0;
    }
    if (option == SPRINTZ_OPT_BLK_KERNELS) {
        if (value < 0 || value > 31) return fail(SPRINTZ_E_INVALID, "SPRINTZ_OPT_BLK_KERNELS is a mask of bits 0 .. 4");
        process().blk_kernels = value;
        return 0;
    }
    if (option == SPRINTZ_OPT_BLK_CHUNKS) {
        if (value < 0 || value > (1ll << 30)) return fail(SPRINTZ_E_INVALID, "SPRINTZ_OPT_BLK_CHUNKS must be in 0..2^30");
        process().blk_chunks = (int)value;
        return 0;
    }
    if (option == SPRINTZ_OPT_LAT_CHUNKS) {
        if (value < 0) return fail(SPRINTZ_E_INVALID, "the batch size must not be negative");
        process().lat_chunks = value;
        return 0;
    }
    if (option == SPRINTZ_OPT_REF_DECODER_QUIRK) { process().ref_quirk = value ? 1 : 0; return 0; }
    if (option == SPRINTZ_OPT_HOST_STREAMS) {
        if (value < 0 || value > 64) return fail(SPRINTZ_E_INVALID, "host streams must be in 0..64");
        process().host_streams = value;                     // (threads that already hold a scratch keep their stream)
        return 0;
    }
    if (option == SPRINTZ_OPT_HOST_WAIT) {
        if (value < 0 || value > 2) return fail(SPRINTZ_E_INVALID, "host wait must be 0 (by callers), 1 (spin) or 2 (sleep)");
        process().host_wait = value;
        return 0;
    }
    if (option == SPRINTZ_OPT_CHUNKS_PER_GROUP) {
        if (value < 1 || value > 64) return fail(SPRINTZ_E_INVALID, "chunks per group must be in 1..64");
        process().chunks_per_group = value;
        return 0;
    }
    return fail(SPRINTZ_E_INVALID, "unknown option");
}
const char* sprintz_mi355x_last_error(void) { return g_last_error.c_str(); }

size_t sprintz_mi355x_compress_bound(int elem_bytes, uint32_t chunk_len, uint16_t ndims)
{
    const size_t esz = (size_t)elem_bytes, D = ndims ? ndims : 1;
    const size_t hb = elem_bytes == 1 ? 3 : 4;
    const size_t hdr_bytes = (2 * D * hb + 7) / 8;
    const size_t max_groups = chunk_len / (16 * D) + 1;
    // header + per group (header + 2 run bytes worst case beyond raw) + raw payload + flush padding
    const size_t b = 8 + max_groups * (hdr_bytes + 3) + (size_t)chunk_len * esz + 32;
    return (b + (SPRINTZ_BOUND_ALIGN - 1)) & ~(size_t)(SPRINTZ_BOUND_ALIGN - 1);
}

uint64_t sprintz_mi355x_num_chunks(uint64_t total_len, uint32_t chunk_len)
{
    if (chunk_len == 0) return 0;
    return (total_len + chunk_len - 1) / chunk_len;
}

int sprintz_mi355x_compress_batch(int codec, int elem_bytes, const void* d_src, uint64_t total_len, uint32_t chunk_len,
                                  uint16_t ndims, void* d_slots, size_t slot_stride, uint32_t* d_sizes, int64_t* d_rets,
                                  void* hip_stream)
{
    int rc = check_common(codec, elem_bytes, ndims);
    if (rc) return rc;
    if (chunk_len == 0 || chunk_len > (1u << 30)) return fail(SPRINTZ_E_INVALID, "chunk_len must be in 1..2^30");
    if (!d_src || !d_slots || !d_sizes) return fail(SPRINTZ_E_INVALID, "null device pointer");
    if (slot_stride % 16 || (uintptr_t)d_slots % 16) return fail(SPRINTZ_E_INVALID, "slots must be 16-byte aligned/strided");
    if (slot_stride < sprintz_mi355x_compress_bound(elem_bytes, chunk_len, ndims))
        return fail(SPRINTZ_E_INVALID, "slot_stride below sprintz_mi355x_compress_bound");
    if ((rc = check_batch_tail(total_len, chunk_len, ndims))) return rc;
    if ((rc = ensure_device())) return rc;
    return encode_launch(codec, elem_bytes, d_src, total_len, chunk_len, ndims, d_slots, slot_stride, d_sizes, d_rets,
                         (hipStream_t)hip_stream, 1);
}

size_t sprintz_mi355x_compress_dense_tmp_bytes(uint64_t nchunks)
{
    // the chained scan's word per workgroup (at most nchunks / 4 workgroups: a chunk takes at most 64 of a workgroup's 256
    // lanes) + the ticket counter -- or the two-launch path's scan scratch, whichever is larger
    const size_t a = (size_t)(nchunks / 4 + 2) * sizeof(uint64_t), b = sprintz_mi355x_compact_tmp_bytes(nchunks);
    return a > b ? a : b;
}

int sprintz_mi355x_compress_batch_dense(int codec, int elem_bytes, const void* d_src, uint64_t total_len, uint32_t chunk_len,
                                        uint16_t ndims, void* d_slots, size_t slot_stride, uint32_t* d_sizes, int64_t* d_rets,
                                        void* d_dense, uint64_t* d_offsets, void* d_tmp, void* hip_stream)
{
    int rc = check_common(codec, elem_bytes, ndims);
    if (rc) return rc;
    if (chunk_len == 0 || chunk_len > (1u << 30)) return fail(SPRINTZ_E_INVALID, "chunk_len must be in 1..2^30");
    if (!d_src || !d_slots || !d_sizes || !d_dense || !d_offsets || !d_tmp) return fail(SPRINTZ_E_INVALID, "null device pointer");
    if (slot_stride % 16 || (uintptr_t)d_slots % 16 || (uintptr_t)d_dense % 16 || (uintptr_t)d_tmp % 8)
        return fail(SPRINTZ_E_INVALID, "slots and the container must be 16-byte aligned/strided");
    if (slot_stride < sprintz_mi355x_compress_bound(elem_bytes, chunk_len, ndims))
        return fail(SPRINTZ_E_INVALID, "slot_stride below sprintz_mi355x_compress_bound");
    if ((rc = check_batch_tail(total_len, chunk_len, ndims))) return rc;
    if ((rc = ensure_device())) return rc;
    hipStream_t st = (hipStream_t)hip_stream;
    const uint64_t nchunks = sprintz_mi355x_num_chunks(total_len, chunk_len);
    if (nchunks == 0) {
        HIP_TRY(hipMemsetAsync(d_offsets, 0, 8, st));
        return 0;
    }
    // (Tried and dropped, measured on the headline batch: the batch in 4 parts, a part's scan + copy on a second stream while
    //  the next part encodes -- 0.87 ms against 0.79 for the launches in a row; the kernels do not fill each other's gaps.)
    const int mode = process().dense_mode.load(std::memory_order_relaxed);
    // chunks too short for a group: all of them verbatim, all sizes known -- written straight into the container (see the kernel)
    if (mode && (codec == SPRINTZ_CODEC_DELTA || codec == SPRINTZ_CODEC_XFF) && !is_lowdim(elem_bytes, ndims) &&
        (chunk_len < 128u || chunk_len < 16u * (uint32_t)ndims) && chunk_len <= 0xffffu) {
        const uint64_t grid = (nchunks * 64 + kThreads - 1) / kThreads;
        if (grid > 0x7fffffffull) return fail(SPRINTZ_E_INVALID, "too many chunks for one launch");
        hipLaunchKernelGGL(verbatim_dense_kernel, dim3((unsigned)grid), dim3(kThreads), 0, st, (const uint8_t*)d_src, total_len, chunk_len,
                           (uint32_t)elem_bytes, (uint32_t)ndims, nchunks, (uint8_t*)d_dense, d_offsets, d_sizes, d_rets);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    DenseRequest dr;
    dr.d_dense = d_dense;
    dr.d_offsets = d_offsets;
    dr.d_tmp = d_tmp;
    rc = encode_launch(codec, elem_bytes, d_src, total_len, chunk_len, ndims, d_slots, slot_stride, d_sizes, d_rets, st, 1, 0, 0, mode ? &dr : nullptr);
    if (rc || dr.fused) return rc;
    // shapes whose encoder has no dense tail (low-dim, more than 64 columns, misaligned blocks): the two-launch path
    return sprintz_mi355x_compact(d_slots, slot_stride, d_sizes, nchunks, 16, d_dense, d_offsets, d_tmp, hip_stream);
}

size_t sprintz_mi355x_compact_tmp_bytes(uint64_t nchunks)
{
    return (size_t)((nchunks + kScanBlock - 1) / kScanBlock + 1) * sizeof(uint64_t);
}

int sprintz_mi355x_compact(const void* d_slots, size_t slot_stride, const uint32_t* d_sizes, uint64_t nchunks, uint32_t align,
                           void* d_dense, uint64_t* d_offsets, void* d_scan_tmp, void* hip_stream)
{
    if (align == 0 || align > 16 || (align & (align - 1))) return fail(SPRINTZ_E_INVALID, "align must be a power of two <= 16");
    if (!d_slots || !d_sizes || !d_dense || !d_offsets || !d_scan_tmp) return fail(SPRINTZ_E_INVALID, "null device pointer");
    int rc = ensure_device();
    if (rc) return rc;
    hipStream_t st = (hipStream_t)hip_stream;
    if (nchunks == 0) {
        HIP_TRY(hipMemsetAsync(d_offsets, 0, 8, st));
        return 0;
    }
    // (round 6: scan AND copy in one launch -- a workgroup per 64 chunks, wave 0 scanning their sizes and finding its place by compact_tail.h's
    //  chained scan, then the waves copying -- was built and measured: BASELINE config 3 at 10 KB 0.291 against 0.271 ms for the whole compress call,
    //  config 1 0.466 against 0.390: the tickets and the look-back cost more than the three ~5 us scan launches they replace.  Not kept.)
    HIP_TRY(launch_size_scan(d_sizes, nchunks, align, d_offsets, d_scan_tmp, st));
    if (slot_stride <= 2048) {
        const uint64_t grid = (nchunks * (1u << SPRINTZ_COPY_SMALL_LOG2) + kThreads - 1) / kThreads;
        hipLaunchKernelGGL(compact_copy_kernel<SPRINTZ_COPY_SMALL_LOG2>, dim3((unsigned)grid), dim3(kThreads), 0, st, (const uint8_t*)d_slots,
                           (uint64_t)slot_stride, d_sizes, d_offsets, nchunks, align, (uint8_t*)d_dense);
    } else {
        const uint64_t grid = (nchunks * 64 + kThreads - 1) / kThreads;
        hipLaunchKernelGGL(compact_copy_kernel<6>, dim3((unsigned)grid), dim3(kThreads), 0, st, (const uint8_t*)d_slots,
                           (uint64_t)slot_stride, d_sizes, d_offsets, nchunks, align, (uint8_t*)d_dense);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int sprintz_mi355x_decompress_batch(int codec, int elem_bytes, const void* d_comp, const uint64_t* d_offsets, uint64_t nchunks,
                                    uint32_t chunk_len, uint16_t ndims, void* d_out, int64_t* d_rets, void* hip_stream)
{
    int rc = check_common(codec, elem_bytes, ndims);
    if (rc) return rc;
    if (chunk_len == 0) return fail(SPRINTZ_E_INVALID, "chunk_len == 0");
    if (!d_comp || !d_offsets || !d_out) return fail(SPRINTZ_E_INVALID, "null device pointer");
    if ((rc = ensure_device())) return rc;
    return decode_launch(codec, elem_bytes, d_comp, d_offsets, nchunks, chunk_len, ndims, d_out, d_rets,
                         (hipStream_t)hip_stream, 0, 0, 0);
}

// ---- drop-in single-call API (host pointers)
int64_t sprintz_mi355x_compress_delta_8b(const uint8_t* s, uint32_t n, int8_t* d, uint16_t nd, int ws)    { return compress_host(SPRINTZ_CODEC_DELTA, 1, s, n, d, nd, ws); }
int64_t sprintz_mi355x_compress_xff_8b(const uint8_t* s, uint32_t n, int8_t* d, uint16_t nd, int ws)      { return compress_host(SPRINTZ_CODEC_XFF, 1, s, n, d, nd, ws); }
int64_t sprintz_mi355x_compress_delta_16b(const uint16_t* s, uint32_t n, int16_t* d, uint16_t nd, int ws) { return compress_host(SPRINTZ_CODEC_DELTA, 2, s, n, d, nd, ws); }
int64_t sprintz_mi355x_compress_xff_16b(const uint16_t* s, uint32_t n, int16_t* d, uint16_t nd, int ws)   { return compress_host(SPRINTZ_CODEC_XFF, 2, s, n, d, nd, ws); }

int64_t sprintz_mi355x_decompress_delta_8b(const int8_t* s, uint8_t* d)    { return decompress_host(SPRINTZ_CODEC_DELTA, 1, s, d, 0, 0, 0, 0); }
int64_t sprintz_mi355x_decompress_xff_8b(const int8_t* s, uint8_t* d)      { return decompress_host(SPRINTZ_CODEC_XFF, 1, s, d, 0, 0, 0, 0); }
int64_t sprintz_mi355x_decompress_delta_16b(const int16_t* s, uint16_t* d) { return decompress_host(SPRINTZ_CODEC_DELTA, 2, s, d, 0, 0, 0, 0); }
int64_t sprintz_mi355x_decompress_xff_16b(const int16_t* s, uint16_t* d)   { return decompress_host(SPRINTZ_CODEC_XFF, 2, s, d, 0, 0, 0, 0); }

int64_t sprintz_mi355x_decompress_noheader(int codec, int elem_bytes, const void* src, void* dest, uint16_t ndims,
                                           uint32_t ngroups, uint16_t remaining_len)
{
    return decompress_host(codec, elem_bytes, src, dest, 1, ndims, ngroups, remaining_len);
}

int64_t sprintz_mi355x_compress_layout(int codec, int elem_bytes, const void* src, uint32_t len, void* dest, uint16_t ndims,
                                       int write_size, int layout)
{
    return compress_host(codec, elem_bytes, src, len, dest, ndims, write_size, layout);
}

int64_t sprintz_mi355x_decompress_layout(int codec, int elem_bytes, const void* src, void* dest, int layout)
{
    return decompress_host(codec, elem_bytes, src, dest, 0, 0, 0, 0, layout);
}

// ---- host convenience: chunked codec over host buffers
int64_t sprintz_mi355x_compress_chunked_host(int codec, int elem_bytes, const void* src, uint64_t total_len, uint32_t chunk_len,
                                             uint16_t ndims, void* comp, size_t comp_capacity, uint64_t* offsets)
{
    int rc = check_common(codec, elem_bytes, ndims);
    if (rc) return rc;
    if (chunk_len == 0) return fail(SPRINTZ_E_INVALID, "chunk_len == 0");
    if ((rc = check_batch_tail(total_len, chunk_len, ndims))) return rc;
    if ((rc = ensure_device())) return rc;
    const uint64_t nchunks = sprintz_mi355x_num_chunks(total_len, chunk_len);
    if (nchunks == 0) { offsets[0] = 0; return 0; }
    const size_t stride = sprintz_mi355x_compress_bound(elem_bytes, chunk_len, ndims);
    DevBuf d_src, d_slots, d_sizes, d_dense, d_offs, d_tmp;
    HIP_TRY(d_src.alloc(total_len * elem_bytes + SPRINTZ_MI355X_READ_SLACK));
    HIP_TRY(d_slots.alloc(stride * nchunks));
    HIP_TRY(d_sizes.alloc(nchunks * 4));
    HIP_TRY(d_offs.alloc((nchunks + 1) * 8));
    HIP_TRY(d_tmp.alloc(sprintz_mi355x_compact_tmp_bytes(nchunks)));
    HIP_TRY(d_dense.alloc(stride * nchunks + SPRINTZ_MI355X_READ_SLACK));
    HIP_TRY(hipMemcpy(d_src.p, src, total_len * elem_bytes, hipMemcpyHostToDevice));
    rc = encode_launch(codec, elem_bytes, d_src.p, total_len, chunk_len, ndims, d_slots.p, stride, (uint32_t*)d_sizes.p,
                       nullptr, nullptr, 1);
    if (rc) return rc;
    rc = sprintz_mi355x_compact(d_slots.p, stride, (const uint32_t*)d_sizes.p, nchunks, 1, d_dense.p, (uint64_t*)d_offs.p,
                                d_tmp.p, nullptr);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(offsets, d_offs.p, (nchunks + 1) * 8, hipMemcpyDeviceToHost));
    const uint64_t total = offsets[nchunks];
    if (total > comp_capacity) return fail(SPRINTZ_E_INVALID, "comp_capacity too small");
    HIP_TRY(hipMemcpy(comp, d_dense.p, total, hipMemcpyDeviceToHost));
    return (int64_t)total;
}

int64_t sprintz_mi355x_decompress_chunked_host(int codec, int elem_bytes, const void* comp, const uint64_t* offsets,
                                               uint64_t nchunks, uint32_t chunk_len, uint16_t ndims, void* out)
{
    int rc = check_common(codec, elem_bytes, ndims);
    if (rc) return rc;
    if (chunk_len == 0) return fail(SPRINTZ_E_INVALID, "chunk_len == 0");
    if ((rc = ensure_device())) return rc;
    if (nchunks == 0) return 0;
    const uint64_t total = offsets[nchunks];
    DevBuf d_comp, d_offs, d_out, d_rets;
    HIP_TRY(d_comp.alloc(total + SPRINTZ_MI355X_READ_SLACK));
    HIP_TRY(d_offs.alloc((nchunks + 1) * 8));
    HIP_TRY(d_out.alloc(nchunks * (uint64_t)chunk_len * elem_bytes));
    HIP_TRY(d_rets.alloc(nchunks * 8));
    HIP_TRY(hipMemcpy(d_comp.p, comp, total, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_offs.p, offsets, (nchunks + 1) * 8, hipMemcpyHostToDevice));
    rc = decode_launch(codec, elem_bytes, d_comp.p, (const uint64_t*)d_offs.p, nchunks, chunk_len, ndims, d_out.p,
                       (int64_t*)d_rets.p, nullptr, 0, 0, 0);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    std::vector<int64_t> rets(nchunks);
    HIP_TRY(hipMemcpy(rets.data(), d_rets.p, nchunks * 8, hipMemcpyDeviceToHost));
    int64_t sum = 0;
    for (uint64_t c = 0; c < nchunks; c++) {
        if (rets[c] < 0) return fail((int)rets[c], "decoder rejected a chunk stream");
        // the copy below takes `sum` CONTIGUOUS elements: every chunk but the last must be full
        if (c + 1 < nchunks && rets[c] != (int64_t)chunk_len)
            return fail(SPRINTZ_E_CORRUPT, "a chunk other than the last decoded to fewer than chunk_len elements");
        sum += rets[c];
    }
    // chunks are full except possibly the last: decoded data is contiguous
    HIP_TRY(hipMemcpy(out, d_out.p, (size_t)sum * elem_bytes, hipMemcpyDeviceToHost));
    return sum;
}

// ---------------------------------------------------------------- query-on-compressed
int sprintz_mi355x_query_batch(int codec, int elem_bytes, const void* d_comp, const uint64_t* d_offsets, uint64_t nchunks,
                               uint32_t chunk_len, uint16_t ndims, int op, int materialize, uint32_t flags, void* d_out,
                               uint64_t* d_partials, int64_t* d_rets, void* hip_stream)
{
    int rc = check_common(codec, elem_bytes, ndims);
    if (rc) return rc;
    if (op < 0 || op > 2) return fail(SPRINTZ_E_INVALID, "op must be 0 (none), 1 (max) or 2 (sum)");
    if (flags & ~(uint32_t)SPRINTZ_QUERY_GENERAL_LAYOUT) return fail(SPRINTZ_E_INVALID, "unknown flag");
    if (chunk_len == 0 || chunk_len > (1u << 30)) return fail(SPRINTZ_E_INVALID, "chunk_len must be in 1..2^30");
    if (!d_comp || !d_offsets) return fail(SPRINTZ_E_INVALID, "null device pointer");
    if (materialize && !d_out) return fail(SPRINTZ_E_INVALID, "materialize without a destination");
    if (op && !d_partials) return fail(SPRINTZ_E_INVALID, "op without a result buffer");
    if ((rc = ensure_device())) return rc;
    QuerySpec qs;
    qs.q = materialize ? (op ? kQueryMaterialize : kQueryOff) : kQueryReduceOnly;
    qs.qop = op;
    qs.qres = op ? d_partials : nullptr;
    qs.general = (flags & SPRINTZ_QUERY_GENERAL_LAYOUT) ? 1 : 0;
    return decode_launch(codec, elem_bytes, d_comp, d_offsets, nchunks, chunk_len, ndims, d_out, d_rets, (hipStream_t)hip_stream,
                         0, 0, 0, qs);
}

int sprintz_mi355x_query_reduce(int op, const uint64_t* d_partials, uint64_t nchunks, uint16_t ndims, uint64_t* d_result,
                                void* hip_stream)
{
    if (op != 1 && op != 2) return fail(SPRINTZ_E_INVALID, "op must be 1 (max) or 2 (sum)");
    if (!d_partials || !d_result || ndims == 0) return fail(SPRINTZ_E_INVALID, "null device pointer");
    int rc = ensure_device();
    if (rc) return rc;
    hipStream_t st = (hipStream_t)hip_stream;
    HIP_TRY(hipMemsetAsync(d_result, 0, (size_t)ndims * 8, st));
    if (nchunks == 0) return 0;
    const uint32_t RB = ndims >= 256 ? 1u : 256u / ndims;
    uint64_t blocks = (nchunks + RB - 1) / RB;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(query_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d_partials, nchunks, (uint32_t)ndims, op,
                       (unsigned long long*)d_result);
    return hipGetLastError() == hipSuccess ? 0 : fail(SPRINTZ_E_HIP, "query_reduce launch");
}

int64_t sprintz_mi355x_query_delta_8b(const int8_t* src, uint8_t* dest, int op, int materialize, uint32_t flags, uint64_t* result)
{
    return query_host(SPRINTZ_CODEC_DELTA, 1, src, dest, op, materialize, result, (flags & SPRINTZ_QUERY_GENERAL_LAYOUT) != 0);
}
int64_t sprintz_mi355x_query_delta_16b(const int16_t* src, uint16_t* dest, int op, int materialize, uint32_t flags, uint64_t* result)
{
    return query_host(SPRINTZ_CODEC_DELTA, 2, src, dest, op, materialize, result, (flags & SPRINTZ_QUERY_GENERAL_LAYOUT) != 0);
}
int64_t sprintz_mi355x_query_xff_8b(const int8_t* src, uint8_t* dest, int op, int materialize, uint32_t flags, uint64_t* result)
{
    return query_host(SPRINTZ_CODEC_XFF, 1, src, dest, op, materialize, result, (flags & SPRINTZ_QUERY_GENERAL_LAYOUT) != 0);
}
int64_t sprintz_mi355x_query_xff_16b(const int16_t* src, uint16_t* dest, int op, int materialize, uint32_t flags, uint64_t* result)
{
    return query_host(SPRINTZ_CODEC_XFF, 2, src, dest, op, materialize, result, (flags & SPRINTZ_QUERY_GENERAL_LAYOUT) != 0);
}

// ---------------------------------------------------------------- column-major matrices (BASELINE config 5)
int sprintz_mi355x_compress_batch_colmajor(int codec, int elem_bytes, const void* d_src, uint64_t nrows, uint64_t col_stride,
                                           uint32_t rows_per_chunk, uint16_t ndims, void* d_slots, size_t slot_stride,
                                           uint32_t* d_sizes, int64_t* d_rets, void* hip_stream)
{
    int rc = check_common(codec, elem_bytes, ndims);
    if (rc) return rc;
    if (rows_per_chunk == 0 || (uint64_t)rows_per_chunk * ndims > (1u << 30))
        return fail(SPRINTZ_E_INVALID, "rows_per_chunk * ndims must be in 1..2^30");
    if (col_stride < nrows) return fail(SPRINTZ_E_INVALID, "col_stride < nrows");
    if (!d_src || !d_slots || !d_sizes) return fail(SPRINTZ_E_INVALID, "null device pointer");
    if (slot_stride % 16 || (uintptr_t)d_slots % 16) return fail(SPRINTZ_E_INVALID, "slots must be 16-byte aligned/strided");
    const uint32_t chunk_len = rows_per_chunk * (uint32_t)ndims;
    if (slot_stride < sprintz_mi355x_compress_bound(elem_bytes, chunk_len, ndims))
        return fail(SPRINTZ_E_INVALID, "slot_stride below sprintz_mi355x_compress_bound");
    if ((rc = ensure_device())) return rc;
    return encode_launch(codec, elem_bytes, d_src, nrows * (uint64_t)ndims, chunk_len, ndims, d_slots, slot_stride, d_sizes, d_rets,
                         (hipStream_t)hip_stream, 1, col_stride);
}

int sprintz_mi355x_compress_batch_colmajor_dense(int codec, int elem_bytes, const void* d_src, uint64_t nrows, uint64_t col_stride,
                                                 uint32_t rows_per_chunk, uint16_t ndims, void* d_slots, size_t slot_stride,
                                                 uint32_t* d_sizes, int64_t* d_rets, void* d_dense, uint64_t* d_offsets, void* d_tmp,
                                                 void* hip_stream)
{
    int rc = check_common(codec, elem_bytes, ndims);
    if (rc) return rc;
    if (rows_per_chunk == 0 || (uint64_t)rows_per_chunk * ndims > (1u << 30))
        return fail(SPRINTZ_E_INVALID, "rows_per_chunk * ndims must be in 1..2^30");
    if (col_stride < nrows) return fail(SPRINTZ_E_INVALID, "col_stride < nrows");
    if (!d_src || !d_slots || !d_sizes || !d_dense || !d_offsets || !d_tmp) return fail(SPRINTZ_E_INVALID, "null device pointer");
    if (slot_stride % 16 || (uintptr_t)d_slots % 16 || (uintptr_t)d_dense % 16 || (uintptr_t)d_tmp % 8)
        return fail(SPRINTZ_E_INVALID, "slots and the container must be 16-byte aligned/strided, d_tmp 8-byte aligned");
    const uint32_t chunk_len = rows_per_chunk * (uint32_t)ndims;
    if (slot_stride < sprintz_mi355x_compress_bound(elem_bytes, chunk_len, ndims))
        return fail(SPRINTZ_E_INVALID, "slot_stride below sprintz_mi355x_compress_bound");
    if ((rc = ensure_device())) return rc;
    hipStream_t st = (hipStream_t)hip_stream;
    const uint64_t total_len = nrows * (uint64_t)ndims, nchunks = sprintz_mi355x_num_chunks(total_len, chunk_len);
    if (nchunks == 0) {
        HIP_TRY(hipMemsetAsync(d_offsets, 0, 8, st));
        return 0;
    }
    DenseRequest dr;
    dr.d_dense = d_dense;
    dr.d_offsets = d_offsets;
    dr.d_tmp = d_tmp;
    rc = encode_launch(codec, elem_bytes, d_src, total_len, chunk_len, ndims, d_slots, slot_stride, d_sizes, d_rets, st, 1, col_stride, 0,
                       process().dense_mode.load(std::memory_order_relaxed) ? &dr : nullptr);
    if (rc || dr.fused) return rc;
    return sprintz_mi355x_compact(d_slots, slot_stride, d_sizes, nchunks, 16, d_dense, d_offsets, d_tmp, hip_stream);
}

int sprintz_mi355x_decompress_batch_colmajor(int codec, int elem_bytes, const void* d_comp, const uint64_t* d_offsets,
                                             uint64_t nchunks, uint32_t rows_per_chunk, uint16_t ndims, uint64_t col_stride,
                                             void* d_out, int64_t* d_rets, void* hip_stream)
{
    int rc = check_common(codec, elem_bytes, ndims);
    if (rc) return rc;
    if (rows_per_chunk == 0 || (uint64_t)rows_per_chunk * ndims > (1u << 30))
        return fail(SPRINTZ_E_INVALID, "rows_per_chunk * ndims must be in 1..2^30");
    if (col_stride < nchunks * (uint64_t)rows_per_chunk) return fail(SPRINTZ_E_INVALID, "col_stride < nchunks * rows_per_chunk");
    if (!d_comp || !d_offsets || !d_out) return fail(SPRINTZ_E_INVALID, "null device pointer");
    if ((rc = ensure_device())) return rc;
    QuerySpec qs;
    qs.col_stride = col_stride;
    return decode_launch(codec, elem_bytes, d_comp, d_offsets, nchunks, rows_per_chunk * (uint32_t)ndims, ndims, d_out, d_rets,
                         (hipStream_t)hip_stream, 0, 0, 0, qs);
}

// ---------------------------------------------------------------- non-RLE codecs, single call (host pointers)
int64_t sprintz_mi355x_compress_norle(int codec, int elem_bytes, const void* src, uint32_t len, void* dest, uint16_t ndims)
{
    if (codec < SPRINTZ_CODEC_DELTA_NORLE || codec > SPRINTZ_CODEC_XFF_NORLE) return fail(SPRINTZ_E_INVALID, "codec must be 2, 3 or 4");
    return compress_host(codec, elem_bytes, src, len, dest, ndims, 1);
}

int64_t sprintz_mi355x_decompress_norle(int codec, int elem_bytes, const void* src, void* dest)
{
    if (codec < SPRINTZ_CODEC_DELTA_NORLE || codec > SPRINTZ_CODEC_XFF_NORLE) return fail(SPRINTZ_E_INVALID, "codec must be 2, 3 or 4");
    return decompress_host(codec, elem_bytes, src, dest, 0, 0, 0, 0);
}

}  // extern "C"

// probe: what does HBM give a kernel that READS r bytes and WRITES w bytes at once, both as coalesced 16-byte-per-lane streams
// (the headline decoder reads 0.47 GB of streams and writes 1.34 GB of samples per launch)?  A fill of 1.34 GB runs at 6.9 TB/s and a
// torch copy at 4.8 TB/s on the same part (tools/probes/hbm_bw.py): the ceiling of a MIXED stream is what the decoder is priced against.
//   ./mix_bw            -> table of (read GB, write GB) -> ms, TB/s;  writes non-temporal like the decoder's
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

// every thread: NL loads then NS stores per trip, all 16 bytes, lane-contiguous; trips stride over the buffers
template <int NL, int NS, bool NT>
__global__ void __launch_bounds__(256) mix(const v4u* __restrict__ src, uint64_t nsrc, v4u* __restrict__ dst, uint64_t ndst)
{
    const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x, nthr = (uint64_t)gridDim.x * 256;
    const uint64_t trips = ndst / ((uint64_t)NS * nthr);
    v4u acc = {1, 2, 3, 4};
    for (uint64_t t = 0; t < trips; t++) {
        v4u in[NL > 0 ? NL : 1];
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const uint64_t i = (t * NL + k) * nthr + tid;      // (nsrc >= ndst and NL <= NS wherever both are used: in range)
            in[k] = __builtin_nontemporal_load(src + i);
        }
#pragma unroll
        for (int k = 0; k < NL; k++) acc ^= in[k];
#pragma unroll
        for (int k = 0; k < NS; k++) {
            const uint64_t i = (t * NS + k) * nthr + tid;
            v4u o = acc;
            o.x += (uint32_t)k;
            if (NT) __builtin_nontemporal_store(o, dst + i); else dst[i] = o;
        }
    }
}

// torch's fill: a workgroup per 16 KB tile, a thread four 16-byte stores 4 KB apart, plain stores, no loop
template <bool NT>
__global__ void __launch_bounds__(256) tile_fill(v4u* __restrict__ dst)
{
    v4u* p = dst + (uint64_t)blockIdx.x * 1024 + threadIdx.x;
    v4u o = {blockIdx.x, threadIdx.x, 3, 4};
#pragma unroll
    for (int k = 0; k < 4; k++) { if (NT) __builtin_nontemporal_store(o, p + 256 * k); else p[256 * k] = o; }
}
// the same tile written by ONE wave per kilobyte line group: thread t of wave w writes 16 B; 16 instructions per thread, 64 KB a workgroup
template <bool NT>
__global__ void __launch_bounds__(256) tile_fill64(v4u* __restrict__ dst)
{
    v4u* p = dst + (uint64_t)blockIdx.x * 4096 + threadIdx.x;
    v4u o = {blockIdx.x, threadIdx.x, 3, 4};
#pragma unroll
    for (int k = 0; k < 16; k++) { if (NT) __builtin_nontemporal_store(o, p + 256 * k); else p[256 * k] = o; }
}
// CALIBRATION (round 4): the guide's 6.29 TB/s is a float4 copy whose workgroups each move ONE tile and leave -- grid = exact
// tiles, NL loads in flight per thread before the first store, no loop.  copy_tile<NL>: 256 threads x NL x 16 B = NL x 4 KB a tile.
template <int NL, bool NT>
__global__ void __launch_bounds__(256) copy_tile(const v4u* __restrict__ src, v4u* __restrict__ dst)
{
    const uint64_t base = (uint64_t)blockIdx.x * (256u * NL) + threadIdx.x;
    v4u in[NL];
#pragma unroll
    for (int k = 0; k < NL; k++) in[k] = __builtin_nontemporal_load(src + base + 256u * k);
#pragma unroll
    for (int k = 0; k < NL; k++) { if (NT) __builtin_nontemporal_store(in[k], dst + base + 256u * k); else dst[base + 256u * k] = in[k]; }
}
// the decoder's mix as tiles: a workgroup reads 6 KB... (NL = 3 pieces of 4 KB x 1/2: see mix_tile) and writes NS x 4 KB, then leaves
template <int NL, int NS, bool NT>
__global__ void __launch_bounds__(256) mix_tile(const v4u* __restrict__ src, v4u* __restrict__ dst)
{
    const uint64_t rb = (uint64_t)blockIdx.x * (256u * NL) + threadIdx.x, wb = (uint64_t)blockIdx.x * (256u * NS) + threadIdx.x;
    v4u acc = {1, 2, 3, 4};
    v4u in[NL];
#pragma unroll
    for (int k = 0; k < NL; k++) in[k] = __builtin_nontemporal_load(src + rb + 256u * k);
#pragma unroll
    for (int k = 0; k < NL; k++) acc ^= in[k];
#pragma unroll
    for (int k = 0; k < NS; k++) { v4u o = acc; o.x += (uint32_t)k; if (NT) __builtin_nontemporal_store(o, dst + wb + 256u * k); else dst[wb + 256u * k] = o; }
}
template <typename K> void run_tiles(K kern, const v4u* src, v4u* dst, uint64_t wbytes, uint64_t rtile, uint64_t wtile, const char* what)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned grid = (unsigned)(wbytes / wtile);
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, src, dst);
    hipEventRecord(e0);
    const int reps = 20;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, src, dst);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double rbs = (double)grid * rtile, wbs = (double)grid * wtile;
    printf("%-28s read %.3f GB + write %.3f GB: %.4f ms -> %.2f TB/s\n", what, rbs / 1e9, wbs / 1e9, ms, (rbs + wbs) / ms / 1e9);
}

template <typename K> void run_fill(K kern, v4u* dst, uint64_t nbytes, uint64_t tile, const char* what)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned grid = (unsigned)(nbytes / tile);
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, dst);
    hipEventRecord(e0);
    const int reps = 20;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, dst);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("%-28s write %.3f GB: %.4f ms -> %.2f TB/s\n", what, (double)grid * tile / 1e9, ms, (double)grid * tile / ms / 1e9);
}

template <int NL, int NS, bool NT = true> void run(const v4u* src, uint64_t nsrc, v4u* dst, uint64_t ndst, const char* what, int grid = 256 * 16)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL((mix<NL, NS, NT>), dim3(grid), dim3(256), 0, 0, src, nsrc, dst, ndst);
    hipEventRecord(e0);
    const int reps = 20;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((mix<NL, NS, NT>), dim3(grid), dim3(256), 0, 0, src, nsrc, dst, ndst);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const uint64_t nthr = (uint64_t)grid * 256, trips = ndst / ((uint64_t)NS * nthr);
    const double wb = (double)trips * NS * nthr * 16, rb = (double)trips * NL * nthr * 16;
    printf("%-28s read %.3f GB + write %.3f GB: %.4f ms -> %.2f TB/s\n", what, rb / 1e9, wb / 1e9, ms, (rb + wb) / ms / 1e9);
}

int main()
{
    const uint64_t wbytes = 1342177280ull, rbytes = 1342177280ull;
    v4u *src, *dst;
    hipMalloc(&src, rbytes); hipMalloc(&dst, wbytes);
    hipMemset(src, 1, rbytes); hipMemset(dst, 0, wbytes);
    const uint64_t ns = rbytes / 16, nd = wbytes / 16;
    run_fill(tile_fill<false>, dst, wbytes, 16384, "tile fill 16 KB, plain");
    run_fill(tile_fill<true>, dst, wbytes, 16384, "tile fill 16 KB, nt");
    run_fill(tile_fill64<false>, dst, wbytes, 65536, "tile fill 64 KB, plain");
    run_fill(tile_fill64<true>, dst, wbytes, 65536, "tile fill 64 KB, nt");
    run<0, 16>(src, ns, dst, nd, "write only, nt");
    run<0, 16, false>(src, ns, dst, nd, "write only, plain");
    run<0, 16, false>(src, ns, dst, nd, "write only, plain, 1024 WGs", 1024);
    run<0, 4, false>(src, ns, dst, nd, "write only, plain, 4/trip", 256 * 64);
    run<6, 17, false>(src, ns, dst, nd, "decoder's mix, plain");
    run<6, 17>(src, ns, dst, nd, "decoder's mix (6 : 17)");
    run<8, 16>(src, ns, dst, nd, "1 : 2");
    run<16, 16>(src, ns, dst, nd, "copy (1 : 1)");
    run<6, 17>(src, ns, dst, nd, "decoder's mix, 16384 WGs", 256 * 64);
    run<6, 17>(src, ns, dst, nd, "decoder's mix, 1024 WGs", 1024);
    run<3, 8>(src, ns, dst, nd, "3 : 8, 16384 WGs", 256 * 64);
    run<16, 0 + 16>(src, ns, dst, nd, "copy again");
    // calibration against the guide's 6.29 TB/s float4 copy: one tile per workgroup, the workgroup leaves after it
    run_tiles(copy_tile<4, false>, src, dst, wbytes, 16384, 16384, "tile copy 16 KB, plain");
    run_tiles(copy_tile<4, true>, src, dst, wbytes, 16384, 16384, "tile copy 16 KB, nt");
    run_tiles(copy_tile<8, false>, src, dst, wbytes, 32768, 32768, "tile copy 32 KB, plain");
    run_tiles(copy_tile<8, true>, src, dst, wbytes, 32768, 32768, "tile copy 32 KB, nt");
    run_tiles(copy_tile<2, false>, src, dst, wbytes, 8192, 8192, "tile copy 8 KB, plain");
    run_tiles(copy_tile<1, false>, src, dst, wbytes, 4096, 4096, "tile copy 4 KB, plain");
    run_tiles(mix_tile<3, 8, false>, src, dst, wbytes, 12288, 32768, "decoder's mix as tiles, plain");
    run_tiles(mix_tile<3, 8, true>, src, dst, wbytes, 12288, 32768, "decoder's mix as tiles, nt");
    run_tiles(mix_tile<6, 17, true>, src, dst, wbytes / 17 * 17 / 69632 * 69632, 24576, 69632, "mix tiles 6:17 (24+68 KB), nt");
    return 0;
}

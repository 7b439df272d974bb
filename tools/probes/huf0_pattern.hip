// probe: the MEMORY PATTERN of the big-batch Huff0 stream kernel without its arithmetic.  3.2 M streams (800 000 chunks x 4), a lane per stream,
// 16 chunks a wavefront, `WAVES` wavefronts a CU (LDS-limited like the kernel); every lane reads its 896-byte stream BACKWARDS in aligned pieces of
// PB bytes (16 bytes a request, all requests of a piece back to back) and its quad writes every stream's 896 output bytes forwards in bursts of OB
// bytes (one stream's OB contiguous bytes per store instruction group, 16 bytes a lane, non-temporal), `pause` x 512 clocks of s_sleep per 64 output
// bytes standing for the symbol chain.  Question: is the stage's 2.4 ms (5.6 GB: 2.3 TB/s) what THIS pattern gets out of the memory system, and
// would 128-byte pieces / bursts change it?     ./huf0_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

template <int PB, int OB, bool RD, bool WR, bool QUAD = false>
__global__ void __launch_bounds__(64) pattern(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint32_t nchunks, uint32_t pause, uint32_t* sink)
{
    extern __shared__ uint8_t lds[];
    const uint32_t t = threadIdx.x, q = t >> 2, j = t & 3;
    const uint64_t chunk = (uint64_t)blockIdx.x * 16 + q;
    if (chunk >= nchunks) return;
    constexpr uint32_t SB = 896;                                  // bytes a stream, in and out
    const uint8_t* ip = in + chunk * (4 * SB) + (uint64_t)j * SB + SB;      // one past the stream's last byte
    uint8_t* op = out + chunk * (4 * SB);                         // the chunk's output; stream s at s * SB
    v4u acc = {0, 0, 0, 0};
    uint32_t in_left = SB;
    for (uint32_t done = 0; done < SB; done += OB) {
        // input: the kernel consumes ~0.95 bytes of stream per output byte: a piece per PB output bytes
        for (uint32_t sub = 0; sub < (OB > PB ? OB / PB : 1); sub++)
        if (RD && done % PB == 0 && in_left >= PB) {
            ip -= PB;
            in_left -= PB;
            if (QUAD) {                                           // the quad fetches stream s's piece together: lane j its bytes 16 j .. (one 64-byte request a quad)
#pragma unroll
                for (int s = 0; s < 4; s++)
#pragma unroll
                    for (int k = 0; k < PB / 64; k++) {
                        const v4u v = __builtin_nontemporal_load((const v4u*)(ip + ((int)s - (int)j) * (int)SB + 64 * k + 16 * j));
                        acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
                    }
            } else {
#pragma unroll
            for (int k = 0; k < PB / 16; k++) {
                const v4u v = __builtin_nontemporal_load((const v4u*)(ip + 16 * k));
                acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
            }
            }
        }
        for (uint32_t p = 0; p < pause * (OB / 64); p++) __builtin_amdgcn_s_sleep(8);     // 8 x 64 clocks
        if (WR) {
#pragma unroll
            for (int s = 0; s < 4; s++)                           // the quad writes stream s's burst: OB / 64 requests of 16 bytes a lane
#pragma unroll
                for (int k = 0; k < OB / 64; k++) {
                    v4u o = {acc.x + s, acc.y, (uint32_t)done, t};
                    __builtin_nontemporal_store(o, (v4u*)(op + s * SB + done + 64 * k + 16 * j));
                }
        }
    }
    if (acc.x == 0x12345678u && sink) sink[0] = acc.y + acc.z + acc.w;
    (void)lds;
}

template <int PB, int OB, bool RD, bool WR, bool QUAD = false> void run(const uint8_t* in, uint8_t* out, uint32_t nchunks, uint32_t pause, int waves)
{
    const unsigned grid = (nchunks + 15) / 16;
    const size_t lds = 160 * 1024 / waves - 512;
    hipFuncSetAttribute(reinterpret_cast<const void*>(pattern<PB, OB, RD, WR, QUAD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((pattern<PB, OB, RD, WR, QUAD>), dim3(grid), dim3(64), lds, 0, in, out, nchunks, pause, nullptr);
    hipEventRecord(e0);
    const int reps = 5;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((pattern<PB, OB, RD, WR, QUAD>), dim3(grid), dim3(64), lds, 0, in, out, nchunks, pause, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double b = (double)nchunks * 4 * 896 * ((RD ? 1 : 0) + (WR ? 1 : 0));
    printf("%s pieces %3d B  bursts %3d B  %s%s  waves/CU %2d  pause %2u: %.3f ms -> %.2f TB/s\n", QUAD ? "quad-wide loads" : "lane loads     ", PB, OB, RD ? "R" : "-", WR ? "W" : "-", waves, pause, ms, b / ms / 1e9);
    fflush(stdout);
}

int main()
{
    const uint32_t nchunks = 800000;
    const uint64_t bytes = (uint64_t)nchunks * 4 * 896;
    uint8_t *in, *out;
    hipMalloc(&in, bytes + 4096); hipMalloc(&out, bytes + 4096);
    hipMemset(in, 1, bytes); hipMemset(out, 0, bytes);
    for (uint32_t pause : {0u, 16u, 24u}) {
        run<64, 64, true, true>(in, out, nchunks, pause, 10);
        run<128, 128, true, true>(in, out, nchunks, pause, 10);
        run<64, 64, true, true, true>(in, out, nchunks, pause, 10);
        run<128, 128, true, true, true>(in, out, nchunks, pause, 10);
        run<128, 64, true, true, true>(in, out, nchunks, pause, 10);
        run<64, 128, true, true, true>(in, out, nchunks, pause, 10);
        run<64, 128, true, true, true>(in, out, nchunks, pause, 8);
        run<128, 128, true, true, true>(in, out, nchunks, pause, 8);
        printf("\n");
    }
    run<64, 64, true, false>(in, out, nchunks, 0, 10);
    run<128, 128, true, false>(in, out, nchunks, 0, 10);
    run<64, 64, true, false, true>(in, out, nchunks, 0, 10);
    run<128, 128, true, false, true>(in, out, nchunks, 0, 10);
    run<64, 64, false, true>(in, out, nchunks, 0, 10);
    run<128, 128, false, true>(in, out, nchunks, 0, 10);
    printf("\n");
    for (int waves : {5, 20}) {
        run<64, 64, true, true>(in, out, nchunks, 8, waves);
        run<64, 64, true, true, true>(in, out, nchunks, 8, waves);
    }
    return 0;
}

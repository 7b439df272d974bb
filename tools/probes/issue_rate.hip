// probe: how fast does ONE wave get its instructions issued, and how does that change with the waves that share its SIMD?
// The statements in DESIGN.md this backs: "a lone wave issues one instruction every ~8 cycles, dependent or not, vector or scalar" (the
// latency of every kernel that leaves the chip empty: decode_lat / encode_lat, the Huff0 leader's tree, a single drop-in call) and "with
// several waves a SIMD the loop is bound by how many instructions a wave has to get through" (the encoders).
// Each wave runs `iters` trips of 256 instructions of one kind and reports its own duration (s_memrealtime, 100 MHz) -- no launch overhead
// in the number.  Waves per SIMD: 1 wave on the whole chip; then k = 1, 2, 4, 8 on every SIMD (256 CUs x 4 SIMDs).
//   dep VALU  : v_alignbit_b32 v, v, v, 7      one chain
//   ind VALU  : eight chains interleaved
//   dep SALU  : s_lshl1_add_u32 s, s, c        one chain
//   ind SALU  : eight scalar chains
//   V/S mixed : independent vector and scalar instructions alternating
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int KIND>
__global__ void __launch_bounds__(1024) probe(uint64_t* __restrict__ ticks, uint32_t* __restrict__ sink, int iters, uint32_t seed)
{
    uint32_t v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = threadIdx.x * 2654435761u + seed + k;
    uint32_t s[8];
#pragma unroll
    for (int k = 0; k < 8; k++) s[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(seed * 31u + k));
    const uint32_t c = seed | 1u;
    const uint32_t cs = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
    __builtin_amdgcn_s_barrier();
    const uint64_t t0 = wall_clock64();
    // (inline asm: the compiler must neither fold a chain nor reorder the kinds)
#define VDEP(r) asm volatile("v_alignbit_b32 %0, %0, %0, 7" : "+v"(r))
#define SDEP(r) asm volatile("s_lshl1_add_u32 %0, %0, %1" : "+s"(r) : "s"(cs) : "scc")
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 32; u++) {
            if constexpr (KIND == 0) {
                VDEP(v[0]); VDEP(v[0]); VDEP(v[0]); VDEP(v[0]); VDEP(v[0]); VDEP(v[0]); VDEP(v[0]); VDEP(v[0]);
            } else if constexpr (KIND == 1) {
                VDEP(v[0]); VDEP(v[1]); VDEP(v[2]); VDEP(v[3]); VDEP(v[4]); VDEP(v[5]); VDEP(v[6]); VDEP(v[7]);
            } else if constexpr (KIND == 2) {
                SDEP(s[0]); SDEP(s[0]); SDEP(s[0]); SDEP(s[0]); SDEP(s[0]); SDEP(s[0]); SDEP(s[0]); SDEP(s[0]);
            } else if constexpr (KIND == 3) {
                SDEP(s[0]); SDEP(s[1]); SDEP(s[2]); SDEP(s[3]); SDEP(s[4]); SDEP(s[5]); SDEP(s[6]); SDEP(s[7]);
            } else {
                VDEP(v[0]); SDEP(s[0]); VDEP(v[1]); SDEP(s[1]); VDEP(v[2]); SDEP(s[2]); VDEP(v[3]); SDEP(s[3]);
            }
        }
    }
    const uint64_t t1 = wall_clock64();
    uint32_t x = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) x ^= v[k] ^ s[k];
    if (x == 0x12345u) sink[0] = x;
    if ((threadIdx.x & 63) == 0) ticks[(uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
static void run(const char* name, double instr_per_trip)
{
    uint64_t* ticks; uint32_t* sink;
    hipMalloc(&ticks, 256 * 2 * 16 * 8); hipMalloc(&sink, 64);
    const int iters = 200;
    struct Cfg { const char* what; int grid, block; } cfgs[] = {
        {"1 wave on the chip", 1, 64}, {"1 wave a SIMD", 256, 256}, {"2 waves a SIMD", 256, 512}, {"4 waves a SIMD", 256, 1024}, {"8 waves a SIMD", 512, 1024}};
    printf("%-10s", name);
    fflush(stdout);
    for (const Cfg& cf : cfgs) {
        hipLaunchKernelGGL(probe<KIND>, dim3(cf.grid), dim3(cf.block), 0, 0, ticks, sink, 3, 7u);
        hipLaunchKernelGGL(probe<KIND>, dim3(cf.grid), dim3(cf.block), 0, 0, ticks, sink, iters, 7u);
        const hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf(" [%s]", hipGetErrorString(e)); fflush(stdout); return; }
        const int nw = cf.grid * (cf.block / 64);
        std::vector<uint64_t> h(nw);
        hipMemcpy(h.data(), ticks, nw * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double ns = (double)h[nw / 2] * 10.0;                       // median wave; 100 MHz ticks
        printf("  %7.2f ns", ns / (iters * instr_per_trip));
        fflush(stdout);
    }
    printf("   per instruction of ONE wave (1 wave on the chip | 1 | 2 | 4 | 8 waves a SIMD)\n");
    hipFree(ticks); hipFree(sink);
}

int main()
{
    printf("time a wave needs per instruction (median wave; 2.4 GHz: 1 ns = 2.4 cycles)\n");
    run<0>("dep VALU", 256);
    run<1>("ind VALU", 256);
    run<2>("dep SALU", 256);
    run<3>("ind SALU", 256);
    run<4>("V/S mixed", 256);
    return 0;
}

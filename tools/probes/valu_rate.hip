// probe: integer VALU issue rate on gfx950 (cycles per wave64 instruction per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, int iters, uint32_t seed)
{
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 + 7, a3 = a0 ^ 5, a4 = a0 + 11, a5 = a0 * 5, a6 = a0 + 13, a7 = a0 ^ 9;
    const uint32_t c = seed | 1;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (OP == 0) { a0 += c; a1 += c; a2 += c; a3 += c; a4 += c; a5 += c; a6 += c; a7 += c; }
            if (OP == 1) { a0 = __mul24(a0, c) ; a1 = __mul24(a1, c); a2 = __mul24(a2, c); a3 = __mul24(a3, c); a4 = __mul24(a4, c); a5 = __mul24(a5, c); a6 = __mul24(a6, c); a7 = __mul24(a7, c); }
            if (OP == 2) { a0 = __builtin_amdgcn_ubfe(a0, c, 9) + 1; a1 = __builtin_amdgcn_ubfe(a1, c, 9) + 1; a2 = __builtin_amdgcn_ubfe(a2, c, 9)+1; a3 = __builtin_amdgcn_ubfe(a3, c, 9)+1; a4 = __builtin_amdgcn_ubfe(a4, c, 9)+1; a5 = __builtin_amdgcn_ubfe(a5, c, 9)+1; a6 = __builtin_amdgcn_ubfe(a6, c, 9)+1; a7 = __builtin_amdgcn_ubfe(a7, c, 9)+1; }
            if (OP == 3) { a0 = (a0 << 3) + a1; a1 = (a1 << 3) + a2; a2 = (a2 << 3) + a3; a3 = (a3 << 3) + a4; a4 = (a4 << 3) + a5; a5 = (a5 << 3) + a6; a6 = (a6 << 3) + a7; a7 = (a7 << 3) + a0; }
            if (OP == 4) { a0 = a0 * c; a1 = a1 * c; a2 = a2 * c; a3 = a3 * c; a4 = a4 * c; a5 = a5 * c; a6 = a6 * c; a7 = a7 * c; }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
template <int OP> void run(const char* name, int instr_per_iter)
{
    uint32_t* d; hipMalloc(&d, 256 * 2048 * 4);
    const int iters = 4000, blocks = 2048;   // 8 blocks per CU = 32 waves/CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 10, 3u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 3u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)blocks * 4 * iters * instr_per_iter;        // wave-instructions
    double per_simd = winstr / 1024.0;
    printf("%-14s %.3f ms  -> %.2f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz), %.2f Tlane-op/s\n", name, ms,
           ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4, winstr * 64 / (ms * 1e-3) / 1e12);
    hipFree(d);
}
int main()
{
    run<0>("v_add_u32", 64);
    run<1>("v_mul_i32_i24", 64);
    run<2>("v_bfe+add", 128);
    run<3>("v_lshl_add", 64);
    run<4>("v_mul_lo_u32", 64);
    return 0;
}

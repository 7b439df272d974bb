// probe: v_mad_i32_i16 with op_sel (high half of src0) and v_add_u32_sdwa src1_sel:WORD_1 on gfx950:
// semantics against the C model, and issue rate.  Used by decode_fast.h's FIRE recurrence.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
__global__ void sem(const int* in, int* out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = in[4 * i], c = in[4 * i + 1], e = in[4 * i + 2];
    int p = in[4 * i + 3], d;
    asm volatile("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(d) : "v"(x), "v"(c), "v"(e));
    asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(p) : "v"(p), "v"(d));
    out[2 * i] = d;
    out[2 * i + 1] = p;
}
__global__ void __launch_bounds__(256) rate(int* out, int iters, int seed)
{
    int a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 + 7, a3 = a0 ^ 5, a4 = a0 + 11, a5 = a0 * 5, a6 = a0 + 13, a7 = a0 ^ 9;
    const int c = seed | 1;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            asm volatile("v_mad_i32_i16 %0, %0, %1, %0 op_sel:[1,0,0,0]" : "+v"(a0) : "v"(c));
            asm volatile("v_mad_i32_i16 %0, %0, %1, %0 op_sel:[1,0,0,0]" : "+v"(a1) : "v"(c));
            asm volatile("v_mad_i32_i16 %0, %0, %1, %0 op_sel:[1,0,0,0]" : "+v"(a2) : "v"(c));
            asm volatile("v_mad_i32_i16 %0, %0, %1, %0 op_sel:[1,0,0,0]" : "+v"(a3) : "v"(c));
            asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(a4) : "v"(a0));
            asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(a5) : "v"(a1));
            asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(a6) : "v"(a2));
            asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(a7) : "v"(a3));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
int main()
{
    const int n = 1 << 16;
    int* h = (int*)malloc(n * 16);
    srand(1);
    for (int i = 0; i < 4 * n; i++) h[i] = (rand() << 16) ^ rand() ^ (rand() << 31);
    int *din, *dout;
    hipMalloc(&din, n * 16); hipMalloc(&dout, n * 8);
    hipMemcpy(din, h, n * 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(sem, dim3(n / 256), dim3(256), 0, 0, din, dout, n);
    int* o = (int*)malloc(n * 8);
    hipMemcpy(o, dout, n * 8, hipMemcpyDeviceToHost);
    int bad_mad = 0, bad_add = 0, low_kept = 0;
    for (int i = 0; i < n; i++) {
        const int x = h[4 * i], c = h[4 * i + 1], e = h[4 * i + 2], p = h[4 * i + 3];
        const int d = (int)((uint32_t)((int)(int16_t)(x >> 16) * (int)(int16_t)c) + (uint32_t)e);
        if (o[2 * i] != d) bad_mad++;
        if ((uint32_t)o[2 * i + 1] != (uint32_t)p + ((uint32_t)d >> 16)) bad_add++;
        (void)low_kept;
    }
    printf("v_mad_i32_i16 op_sel:[1,0,0,0]: %d mismatches of %d; v_add_u32_sdwa src1_sel:WORD_1: %d mismatches (%d)\n",
           bad_mad, n, bad_add, low_kept);
    int* d2; hipMalloc(&d2, 256 * 2048 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate, dim3(2048), dim3(256), 0, 0, d2, 10, 3);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate, dim3(2048), dim3(256), 0, 0, d2, 4000, 3);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double winstr = 2048.0 * 4 * 4000 * 64;
    printf("mad_i32_i16 + add_u16 mix: %.3f ms -> %.2f cycles per wave-instr per SIMD @2.4GHz\n", ms, ms * 1e6 / (winstr / 1024.0) * 2.4);
    return 0;
}

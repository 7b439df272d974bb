// probe: does ds_read_b32 at an unaligned LDS byte address return the right bytes on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void k(uint32_t* out)
{
    __shared__ __attribute__((aligned(16))) uint8_t s[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) s[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    typedef uint32_t __attribute__((aligned(1), may_alias)) u32_u;
    uint32_t off = threadIdx.x * 5 + 1;          // all residues mod 4
    out[threadIdx.x] = *(const u32_u*)(s + off);
    typedef uint16_t __attribute__((aligned(1), may_alias)) u16_u;
    out[64 + threadIdx.x] = *(const u16_u*)(s + off);
    // forced single instruction
    uint32_t v;
    uint32_t addr = (uint32_t)(uintptr_t)(s + off);
    asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[128 + threadIdx.x] = v;
}
int main()
{
    uint32_t* d; hipMalloc(&d, 192 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    std::vector<uint32_t> h(192); hipMemcpy(h.data(), d, 192 * 4, hipMemcpyDeviceToHost);
    int bad = 0, bad16 = 0, badasm = 0;
    for (int t = 0; t < 64; t++) {
        uint32_t off = t * 5 + 1, want = 0;
        for (int b = 0; b < 4; b++) want |= (uint32_t)(uint8_t)((off + b) * 7 + 3) << (8 * b);
        if (h[t] != want) bad++;
        if (h[64 + t] != (want & 0xffff)) bad16++;
        if (h[128 + t] != want) badasm++;
    }
    printf("unaligned LDS: compiler-load bad=%d  u16 bad=%d  raw ds_read_b32 bad=%d\n", bad, bad16, badasm);
    return 0;
}

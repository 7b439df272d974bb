// probe: what does ds_read2_b32 do with a byte address that is not a multiple of 4 on gfx950 --
// (a) which bytes come back, (b) what does it cost next to the aligned form?  (decode_fast.h spends
// one v_and_b32 per sample on aligning the address it hands to ds_read2_b32.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void semantics(uint32_t* out)
{
    __shared__ __attribute__((aligned(16))) uint8_t s[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) s[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    const uint32_t addr = (uint32_t)(uintptr_t)s + threadIdx.x * 5 + 1;
    uint64_t v;
    asm volatile("ds_read2_b32 %0, %1 offset1:1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[threadIdx.x] = (uint32_t)v;
    out[64 + threadIdx.x] = (uint32_t)(v >> 32);
    uint64_t w;
    asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(addr) : "memory");
    out[128 + threadIdx.x] = (uint32_t)w;
    out[192 + threadIdx.x] = (uint32_t)(w >> 32);
}
template <int MODE>
__global__ void __launch_bounds__(256) timing(uint32_t* out, int iters, uint32_t phase)
{
    __shared__ __attribute__((aligned(16))) uint8_t s[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) s[i] = (uint8_t)i;
    __syncthreads();
    uint32_t addr = (uint32_t)(uintptr_t)s + (threadIdx.x * 36u) % 8192u + phase;
    uint32_t acc = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            uint64_t v;
            if (MODE == 0) {          // aligned: and + ds_read2_b32
                uint32_t q = (addr + u * 20) & ~3u;
                asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(v) : "v"(q) : "memory");
            } else if (MODE == 1) {   // raw address into ds_read2_b32
                uint32_t q = addr + u * 20;
                asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(v) : "v"(q) : "memory");
            } else {                  // ds_read_b64 raw
                uint32_t q = addr + u * 20;
                asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(q) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            acc += (uint32_t)v ^ (uint32_t)(v >> 32);
        }
        addr += acc & 4u;   // keep the phase (adds 0 or 4)
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int MODE> float run(uint32_t* d, uint32_t phase)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(timing<MODE>, dim3(2048), dim3(256), 0, 0, d, 10, phase);
    hipEventRecord(e0);
    hipLaunchKernelGGL(timing<MODE>, dim3(2048), dim3(256), 0, 0, d, 2000, phase);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main()
{
    uint32_t* d; hipMalloc(&d, 2048 * 256 * 4);
    hipLaunchKernelGGL(semantics, dim3(1), dim3(64), 0, 0, d);
    std::vector<uint32_t> h(256); hipMemcpy(h.data(), d, 256 * 4, hipMemcpyDeviceToHost);
    int exact = 0, dropped = 0, b64_exact = 0;
    for (int t = 0; t < 64; t++) {
        const uint32_t off = t * 5 + 1;
        auto bytes = [&](uint32_t o) { uint32_t w = 0; for (int b = 0; b < 4; b++) w |= (uint32_t)(uint8_t)((o + b) * 7 + 3) << (8 * b); return w; };
        if (h[t] == bytes(off) && h[64 + t] == bytes(off + 4)) exact++;
        if (h[t] == bytes(off & ~3u) && h[64 + t] == bytes((off & ~3u) + 4)) dropped++;
        if (h[128 + t] == bytes(off) && h[192 + t] == bytes(off + 4)) b64_exact++;
    }
    printf("ds_read2_b32 at a misaligned address: byte-exact in %d/64 lanes, low-2-bits-dropped in %d/64 lanes (16 lanes are aligned anyway)\n", exact, dropped);
    printf("ds_read_b64  at a misaligned address: byte-exact in %d/64 lanes\n", b64_exact);
    for (uint32_t phase = 0; phase < 4; phase++)
        printf("phase %u: and+ds_read2_b32 %.3f ms | raw ds_read2_b32 %.3f ms | raw ds_read_b64 %.3f ms\n", phase, run<0>(d, phase), run<1>(d, phase), run<2>(d, phase));
    return 0;
}

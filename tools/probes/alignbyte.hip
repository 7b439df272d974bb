// probe: does v_alignbyte_b32 use S2[1:0] or S2[4:0] on gfx950?  And ds_read2_b32 needs 4-byte alignment?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint32_t* out)
{
    // operands come from memory so nothing can be constant-folded
    uint32_t lo = out[32], hi = out[33];
    out[threadIdx.x] = __builtin_amdgcn_alignbyte(hi, lo, out[40 + threadIdx.x]);
}
int main()
{
    uint32_t* d; (void)hipMalloc(&d, 256);
    uint32_t init[64] = {0}; init[32] = 0x44332211u; init[33] = 0x88776655u; for (int i = 0; i < 8; i++) init[40 + i] = i;
    (void)hipMemcpy(d, init, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(8), 0, 0, d);
    uint32_t h[8]; (void)hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    for (int s = 0; s < 8; s++) printf("alignbyte(hi,lo,%d) = %08x\n", s, h[s]);
    return 0;
}

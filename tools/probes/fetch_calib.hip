// probe: what does rocprofv3's FETCH_SIZE report for reads whose TRUE size is known?  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE counts a wide
// coalesced read stream's 128-byte requests at 64 bytes (hence "x 2").  The univariate decoder (decode_uni.h) does not read like that: every
// lane fetches 64-byte bursts of its OWN stream, ~440 bytes apart from its neighbour's.  Three kernels, each reading exactly 512 MiB once:
//   wide   : a wave reads 1 KB contiguous per instruction (16 bytes a lane, lanes adjacent)         -> the pattern the correction was made for
//   burst64: lane l reads the 64 contiguous bytes at l * 448 + 64 * k (four 16-byte loads), k = 0 .. 6  -> decode_uni's refill
//   burst64x2: the same, but a lane's two 64-byte halves of a 128-byte line are read 3 bursts apart   -> does the second half come from HBM again?
// run under:  rocprofv3 --pmc FETCH_SIZE --kernel-trace ...   and compare KiB with 524288.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) wide(const v4u* __restrict__ src, uint32_t* __restrict__ sink, uint64_t n16)
{
    v4u acc = {0, 0, 0, 0};
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) acc ^= src[i];
    if (acc.x == 0x12345678u) sink[0] = acc.y;
}
// lane = one "stream" of `per` bytes (a multiple of 64); streams are contiguous in memory; every trip a lane reads its next 64 bytes
template <int ORDER>
__global__ void __launch_bounds__(256) burst64(const uint8_t* __restrict__ src, uint32_t* __restrict__ sink, uint64_t nstreams, uint32_t per)
{
    const uint64_t l = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (l >= nstreams) return;
    const uint8_t* const s = src + l * per;
    v4u acc = {0, 0, 0, 0};
    const uint32_t nb = per / 64;
    for (uint32_t k = 0; k < nb; k++) {
        // ORDER 0: bursts in address order; ORDER 1: even bursts first, then the odd ones (a line's halves far apart in time)
        const uint32_t kk = ORDER == 0 ? k : (k < (nb + 1) / 2 ? 2 * k : 2 * (k - (nb + 1) / 2) + 1);
        const v4u* p = (const v4u*)(s + 64u * kk);
        acc ^= p[0] ^ p[1] ^ p[2] ^ p[3];
    }
    if (acc.x == 0x12345678u) sink[0] = acc.y;
}

int main()
{
    const uint64_t bytes = 512ull << 20;
    uint8_t* src; uint32_t* sink;
    hipMalloc(&src, bytes + 4096); hipMalloc(&sink, 64);
    hipMemset(src, 1, bytes + 4096);
    const uint32_t per = 448;                                  // a cfg1 stream is ~440 bytes
    const uint64_t nstreams = bytes / per;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(wide, dim3(4096), dim3(256), 0, 0, (const v4u*)src, sink, bytes / 16);
        hipLaunchKernelGGL(burst64<0>, dim3((unsigned)((nstreams + 255) / 256)), dim3(256), 0, 0, src, sink, nstreams, per);
        hipLaunchKernelGGL(burst64<1>, dim3((unsigned)((nstreams + 255) / 256)), dim3(256), 0, 0, src, sink, nstreams, per);
    }
    hipDeviceSynchronize();
    printf("each kernel read %llu KiB (wide) / %llu KiB (bursts)\n", (unsigned long long)(bytes >> 10), (unsigned long long)((nstreams * per) >> 10));
    return 0;
}

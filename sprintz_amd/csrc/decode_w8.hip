// decode_w8.hip -- instantiations of the batched decoder for 8-bit elements.
#include "launch.h"
namespace sprintz {
hipError_t launch_decode_w8(bool fire, bool lowdim, int cpl, int q, unsigned grid, size_t shmem, hipStream_t st, const DecodeArgs& a)
{
    SPRINTZ_DISPATCH(decode_kernel, 8)
}
hipError_t launch_decode_fast_w8(bool fire, int dp, int cpl, bool exact, int q, unsigned grid, size_t shmem, hipStream_t st, const DecodeArgs& a)
{
    SPRINTZ_DISPATCH_DECODE_FAST(decode_fast_kernel, 8)
}
hipError_t launch_decode_uni_w8(bool fire, unsigned grid, hipStream_t st, const DecodeArgs& a)
{
    if (fire) hipLaunchKernelGGL((decode_uni_kernel<8, true>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((decode_uni_kernel<8, false>), dim3(grid), dim3(256), 0, st, a);
    return hipGetLastError();
}
}  // namespace sprintz

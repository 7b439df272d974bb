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
hipError_t launch_decode_uni_w8(bool fire, int nd, unsigned grid, hipStream_t st, const DecodeArgs& a)
{
    switch (nd) {
        case 1:
            if (fire) hipLaunchKernelGGL((decode_uni_kernel<8, true, 1>), dim3(grid), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((decode_uni_kernel<8, false, 1>), dim3(grid), dim3(256), 0, st, a);
            break;
        case 2:
            if (fire) hipLaunchKernelGGL((decode_uni_kernel<8, true, 2>), dim3(grid), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((decode_uni_kernel<8, false, 2>), dim3(grid), dim3(256), 0, st, a);
            break;
        case 3:
            if (fire) hipLaunchKernelGGL((decode_uni_kernel<8, true, 3>), dim3(grid), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((decode_uni_kernel<8, false, 3>), dim3(grid), dim3(256), 0, st, a);
            break;
        case 4:
            if (fire) hipLaunchKernelGGL((decode_uni_kernel<8, true, 4>), dim3(grid), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((decode_uni_kernel<8, false, 4>), dim3(grid), dim3(256), 0, st, a);
            break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
}  // namespace sprintz

// encode_w16.hip -- instantiations of the batched encoder for 16-bit elements.
#include "launch.h"
namespace sprintz {
hipError_t launch_encode_w16(bool fire, bool lowdim, int cpl, unsigned grid, size_t shmem, hipStream_t st, const EncodeArgs& a)
{
    SPRINTZ_DISPATCH_ENC(encode_kernel, 16)
}
hipError_t launch_encode_fast_w16(bool fire, int dp, bool exact, unsigned grid, size_t shmem, hipStream_t st, const EncodeArgs& a)
{
    SPRINTZ_DISPATCH_FAST(encode_fast_kernel, 16)
}
hipError_t launch_encode_uni_w16(bool fire, int nd, unsigned grid, hipStream_t st, const EncodeArgs& a)
{
    switch (nd) {
        case 1:
            if (fire) hipLaunchKernelGGL((encode_uni_kernel<16, true, 1>), dim3(grid), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((encode_uni_kernel<16, false, 1>), dim3(grid), dim3(256), 0, st, a);
            break;
        case 2:
            if (fire) hipLaunchKernelGGL((encode_uni_kernel<16, true, 2>), dim3(grid), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((encode_uni_kernel<16, false, 2>), dim3(grid), dim3(256), 0, st, a);
            break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
#define SPRINTZ_PAIR_CASE(DPV)                                                                                                    \
    case DPV:                                                                                                                      \
        if (a.col_stride) {                                                                                                        \
            if (exact) return fire ? launch_one(encode_wide_kernel<16, true, true, false, DPV, true>, grid, shmem, st, a)          \
                                   : launch_one(encode_wide_kernel<16, false, true, false, DPV, true>, grid, shmem, st, a);        \
            return fire ? launch_one(encode_wide_kernel<16, true, false, false, DPV, true>, grid, shmem, st, a)                    \
                        : launch_one(encode_wide_kernel<16, false, false, false, DPV, true>, grid, shmem, st, a);                  \
        }                                                                                                                          \
        if (exact) return fire ? launch_one(encode_wide_kernel<16, true, true, false, DPV>, grid, shmem, st, a)                    \
                               : launch_one(encode_wide_kernel<16, false, true, false, DPV>, grid, shmem, st, a);                  \
        return fire ? launch_one(encode_wide_kernel<16, true, false, false, DPV>, grid, shmem, st, a)                              \
                    : launch_one(encode_wide_kernel<16, false, false, false, DPV>, grid, shmem, st, a);
hipError_t launch_encode_pair_w16(bool fire, int dp, bool exact, unsigned grid, size_t shmem, hipStream_t st, const EncodeArgs& a)
{
    switch (dp) {
        SPRINTZ_PAIR_CASE(4)
        SPRINTZ_PAIR_CASE(8)
        SPRINTZ_PAIR_CASE(16)
        SPRINTZ_PAIR_CASE(32)
        default: return hipErrorInvalidValue;
    }
}
#undef SPRINTZ_PAIR_CASE
hipError_t launch_encode_wide_w16(bool fire, bool exact, unsigned grid, size_t shmem, hipStream_t st, const EncodeArgs& a)
{
    if (exact) return fire ? launch_one(encode_wide_kernel<16, true, true>, grid, shmem, st, a) : launch_one(encode_wide_kernel<16, false, true>, grid, shmem, st, a);
    return fire ? launch_one(encode_wide_kernel<16, true, false>, grid, shmem, st, a) : launch_one(encode_wide_kernel<16, false, false>, grid, shmem, st, a);
}
}  // namespace sprintz

// decode_w16.hip -- instantiations of the batched decoder for 16-bit elements.
#include "launch.h"
namespace sprintz {
hipError_t launch_decode_w16(bool fire, bool lowdim, int cpl, int q, unsigned grid, size_t shmem, hipStream_t st, const DecodeArgs& a)
{
    SPRINTZ_DISPATCH(decode_kernel, 16)
}
hipError_t launch_decode_fast_w16(bool fire, int dp, int cpl, bool exact, int q, int ds, unsigned grid, size_t shmem, hipStream_t st, const DecodeArgs& a)
{
    if (ds == 80) {                                        // 65 .. 80 columns on 64 x 2 with the LDS carve of 80 columns: 12 waves a CU instead of 8
        if (dp != 64 || cpl != 2 || q != kQueryOff || a.col_stride || exact || a.D <= 64 || a.D > 80) return hipErrorInvalidValue;
        return fire ? launch_one(decode_fast_kernel<16, true, 64, 2, false, kQueryOff, false, 80>, grid, shmem, st, a)
                    : launch_one(decode_fast_kernel<16, false, 64, 2, false, kQueryOff, false, 80>, grid, shmem, st, a);
    }
    if (ds != 0) return hipErrorInvalidValue;
    SPRINTZ_DISPATCH_DECODE_FAST(decode_fast_kernel, 16)
}
#define SPRINTZ_UNI_CASE(NDV, QV)                                                                          \
    if (nd == NDV && q == QV) {                                                                             \
        constexpr int tpb = decode_uni_threads(16, NDV);                                                   \
        const unsigned g = (unsigned)((a.nchunks + tpb - 1) / tpb);                                          \
        if (fire) hipLaunchKernelGGL((decode_uni_kernel<16, true, NDV, QV>), dim3(g), dim3(tpb), 0, st, a);   \
        else hipLaunchKernelGGL((decode_uni_kernel<16, false, NDV, QV>), dim3(g), dim3(tpb), 0, st, a);       \
        return hipGetLastError();                                                                           \
    }
hipError_t launch_decode_uni_w16(bool fire, int nd, int q, unsigned grid, hipStream_t st, const DecodeArgs& a)
{
    SPRINTZ_UNI_CASE(1, kQueryOff)
    SPRINTZ_UNI_CASE(1, kQueryMaterialize)
    SPRINTZ_UNI_CASE(1, kQueryReduceOnly)
    SPRINTZ_UNI_CASE(2, kQueryOff)
    SPRINTZ_UNI_CASE(2, kQueryMaterialize)
    SPRINTZ_UNI_CASE(2, kQueryReduceOnly)
    return hipErrorInvalidValue;
}
#undef SPRINTZ_UNI_CASE
}  // namespace sprintz

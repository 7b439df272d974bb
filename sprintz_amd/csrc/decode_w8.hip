// decode_w8.hip -- instantiations of the batched decoder for 8-bit elements.
#include "launch.h"
namespace sprintz {
hipError_t launch_decode_w8(bool fire, bool lowdim, int cpl, int q, unsigned grid, size_t shmem, hipStream_t st, const DecodeArgs& a)
{
    SPRINTZ_DISPATCH(decode_kernel, 8)
}
hipError_t launch_decode_fast_w8(bool fire, int dp, int cpl, bool exact, int q, int ds, unsigned grid, size_t shmem, hipStream_t st, const DecodeArgs& a)
{
    if (dp == 32 && cpl == 3) {                            // the split mapping: 8 bits, 65 .. 80 columns, plain row-major decode only
        if (ds != 80 || q != kQueryOff || a.col_stride || exact || a.D <= 64 || a.D > 80) return hipErrorInvalidValue;
        return fire ? launch_one(decode_fast_kernel<8, true, 32, 3, false, kQueryOff, false, 80>, grid, shmem, st, a)
                    : launch_one(decode_fast_kernel<8, false, 32, 3, false, kQueryOff, false, 80>, grid, shmem, st, a);
    }
    if (ds != 0) return hipErrorInvalidValue;
    SPRINTZ_DISPATCH_DECODE_FAST(decode_fast_kernel, 8)
}
#define SPRINTZ_UNI_CASE(NDV, QV)                                                                          \
    if (nd == NDV && q == QV) {                                                                             \
        constexpr int tpb = decode_uni_threads(8, NDV);                                                   \
        const unsigned g = (unsigned)((a.nchunks + tpb - 1) / tpb);                                          \
        if (fire) hipLaunchKernelGGL((decode_uni_kernel<8, true, NDV, QV>), dim3(g), dim3(tpb), 0, st, a);   \
        else hipLaunchKernelGGL((decode_uni_kernel<8, false, NDV, QV>), dim3(g), dim3(tpb), 0, st, a);       \
        return hipGetLastError();                                                                           \
    }
hipError_t launch_decode_uni_w8(bool fire, int nd, int q, unsigned grid, hipStream_t st, const DecodeArgs& a)
{
    SPRINTZ_UNI_CASE(1, kQueryOff)
    SPRINTZ_UNI_CASE(1, kQueryMaterialize)
    SPRINTZ_UNI_CASE(1, kQueryReduceOnly)
    SPRINTZ_UNI_CASE(2, kQueryOff)
    SPRINTZ_UNI_CASE(2, kQueryMaterialize)
    SPRINTZ_UNI_CASE(2, kQueryReduceOnly)
    SPRINTZ_UNI_CASE(3, kQueryOff)
    SPRINTZ_UNI_CASE(3, kQueryMaterialize)
    SPRINTZ_UNI_CASE(3, kQueryReduceOnly)
    SPRINTZ_UNI_CASE(4, kQueryOff)
    SPRINTZ_UNI_CASE(4, kQueryMaterialize)
    SPRINTZ_UNI_CASE(4, kQueryReduceOnly)
    return hipErrorInvalidValue;
}
#undef SPRINTZ_UNI_CASE
}  // namespace sprintz

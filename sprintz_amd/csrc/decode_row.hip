// decode_row.hip -- instantiations of the piece-sequential delta decoder (decode_row.h).
#include "launch.h"
#include "decode_row.h"
namespace sprintz {
hipError_t launch_decode_row(int w, unsigned grid, hipStream_t st, const DecodeArgs& a, const RowDecGeom& g)
{
    if (!g.ok) return hipErrorInvalidValue;
    if (w == 8) hipLaunchKernelGGL(decode_row_kernel<8>, dim3(grid), dim3(256), 0, st, a, g);
    else hipLaunchKernelGGL(decode_row_kernel<16>, dim3(grid), dim3(256), 0, st, a, g);
    return hipGetLastError();
}
}  // namespace sprintz

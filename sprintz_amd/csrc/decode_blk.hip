// decode_blk.hip -- instantiations of the block-parallel delta decoder (decode_blk.h).
#include "launch.h"
#include "decode_blk.h"
namespace sprintz {
hipError_t launch_decode_blk(int w, unsigned grid, hipStream_t st, const DecodeArgs& a, const BlkDecGeom& g)
{
    if (!g.ok) return hipErrorInvalidValue;
    if (w == 8) return launch_with_lds(decode_blk_kernel<8>, grid, 256u, g.total, st, a, g);
    return launch_with_lds(decode_blk_kernel<16>, grid, 256u, g.total, st, a, g);
}
}  // namespace sprintz

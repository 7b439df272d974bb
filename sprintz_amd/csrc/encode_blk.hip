// encode_blk.hip -- instantiations of the block-parallel delta encoder (encode_blk.h).
#include "launch.h"
#include "encode_blk.h"
namespace sprintz {
hipError_t launch_encode_blk(int w, unsigned grid, hipStream_t st, const EncodeArgs& a, const BlkEncGeom& g)
{
    if (!g.ok) return hipErrorInvalidValue;
    if (w == 8) return launch_with_lds(encode_blk_kernel<8>, grid, 256u, g.total, st, a, g);
    return launch_with_lds(encode_blk_kernel<16>, grid, 256u, g.total, st, a, g);
}
hipError_t launch_encode_blk_uni(int w, unsigned grid, hipStream_t st, const EncodeArgs& a, const BlkEncGeom& g)
{
    if (!g.ok) return hipErrorInvalidValue;
    if (w == 8) return launch_with_lds(encode_blk_uni_kernel<8>, grid, 256u, g.total, st, a, g);
    return launch_with_lds(encode_blk_uni_kernel<16>, grid, 256u, g.total, st, a, g);
}
}  // namespace sprintz

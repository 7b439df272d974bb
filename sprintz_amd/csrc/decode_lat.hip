// decode_lat.hip -- instantiations of the latency decoder (decode_lat.h): one workgroup per chunk.
#include "launch.h"
#include "decode_lat.h"
namespace sprintz {
namespace {
template <typename K>
hipError_t launch_lat(K kernel, unsigned grid, const LatCarve& c, hipStream_t st, const DecodeArgs& a)
{
    return launch_with_lds(kernel, grid, 256u, c.total, st, a, c);      // (> 48 KB: the attribute once per instantiation and device, lds_attr.h)
}
}  // namespace
#define SPRINTZ_LAT_CASE(WV, DPV)                                                                         \
    if (w == WV && dp == DPV)                                                                             \
        return fire ? launch_lat(decode_lat_kernel<WV, true, DPV>, grid, c, st, a)                        \
                    : launch_lat(decode_lat_kernel<WV, false, DPV>, grid, c, st, a);
hipError_t launch_decode_lat(int w, bool fire, int dp, bool lowdim, unsigned grid, uint32_t bound_bytes, hipStream_t st, const DecodeArgs& a)
{
    const LatCarve c = lat_carve(bound_bytes, a.chunk_len, (uint32_t)a.D);
    if (c.total > 150 * 1024) return hipErrorInvalidValue;
    if (lowdim) {                                          // D <= 4 at 8 bits, <= 2 at 16: four lanes a group
        if (dp != 4) return hipErrorInvalidValue;
        if (w == 8) return fire ? launch_lat(decode_lat_kernel<8, true, 4, true>, grid, c, st, a) : launch_lat(decode_lat_kernel<8, false, 4, true>, grid, c, st, a);
        return fire ? launch_lat(decode_lat_kernel<16, true, 4, true>, grid, c, st, a) : launch_lat(decode_lat_kernel<16, false, 4, true>, grid, c, st, a);
    }
    SPRINTZ_LAT_CASE(16, 4)
    SPRINTZ_LAT_CASE(16, 8)
    SPRINTZ_LAT_CASE(16, 16)
    SPRINTZ_LAT_CASE(16, 32)
    SPRINTZ_LAT_CASE(16, 64)
    SPRINTZ_LAT_CASE(8, 8)
    SPRINTZ_LAT_CASE(8, 16)
    SPRINTZ_LAT_CASE(8, 32)
    SPRINTZ_LAT_CASE(8, 64)
    return hipErrorInvalidValue;
}
#undef SPRINTZ_LAT_CASE
}  // namespace sprintz

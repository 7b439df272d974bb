// group_ops.h -- scans/reductions inside a DP-lane group of one wavefront with
// DPP (data-parallel primitives: the cross-lane operand modifiers of VOP1/VOP2
// on gfx9-class hardware).  No LDS, no ds_bpermute for DP <= 16.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sprintz {

// dpp_ctrl encodings (AMDGPU ISA)
constexpr int DPP_QUAD_PERM(int a, int b, int c, int d) { return a | (b << 2) | (c << 4) | (d << 6); }
constexpr int DPP_ROW_SHR(int n) { return 0x110 + n; }
constexpr int DPP_ROW_SHL(int n) { return 0x100 + n; }
constexpr int DPP_ROW_HALF_MIRROR = 0x141;
constexpr int DPP_ROW_MIRROR = 0x140;

// value of the lane selected by CTRL; lanes whose source is outside the 16-lane
// row read 0 (bound_ctrl); lanes outside BANK_MASK keep `old`
template <int CTRL, int BANK_MASK = 0xf>
__device__ __forceinline__ uint32_t dpp(uint32_t old, uint32_t src)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, 0xf, BANK_MASK, true);
}

// Exclusive prefix sum and total over groups of DP adjacent lanes (DP = 4, 8, 16
// via DPP; 32, 64 finish with wave shuffles).  x must be small enough that the
// group total fits 32 bits.  lane_d = lane index inside the group.
template <int DP>
__device__ __forceinline__ uint32_t group_scan(uint32_t x, int lane_d, uint32_t& total)
{
    uint32_t incl = x;
    if constexpr (DP == 4) {
        // the DPP moves run unconditionally (a DPP read of an EXEC-disabled lane yields 0);
        // only the accumulate is predicated on the lane's position in its group
        const uint32_t s1 = dpp<DPP_ROW_SHR(1)>(0, incl);
        incl += (lane_d >= 1) ? s1 : 0u;
        const uint32_t s2 = dpp<DPP_ROW_SHR(2)>(0, incl);
        incl += (lane_d >= 2) ? s2 : 0u;
        uint32_t t = x + dpp<DPP_QUAD_PERM(1, 0, 3, 2)>(0, x);
        total = t + dpp<DPP_QUAD_PERM(2, 3, 0, 1)>(0, t);
    } else if constexpr (DP == 8) {
        const uint32_t s1 = dpp<DPP_ROW_SHR(1)>(0, incl);
        incl += (lane_d >= 1) ? s1 : 0u;
        const uint32_t s2 = dpp<DPP_ROW_SHR(2)>(0, incl);
        incl += (lane_d >= 2) ? s2 : 0u;
        incl += dpp<DPP_ROW_SHR(4), 0xa>(0, incl);                  // banks 1,3 = lanes 4-7 of each group
        uint32_t t = x + dpp<DPP_QUAD_PERM(1, 0, 3, 2)>(0, x);
        t += dpp<DPP_QUAD_PERM(2, 3, 0, 1)>(0, t);
        total = t + dpp<DPP_ROW_HALF_MIRROR>(0, t);
    } else if constexpr (DP == 16) {
        incl += dpp<DPP_ROW_SHR(1)>(0, incl);
        incl += dpp<DPP_ROW_SHR(2)>(0, incl);
        incl += dpp<DPP_ROW_SHR(4)>(0, incl);
        incl += dpp<DPP_ROW_SHR(8)>(0, incl);
        uint32_t t = x + dpp<DPP_QUAD_PERM(1, 0, 3, 2)>(0, x);
        t += dpp<DPP_QUAD_PERM(2, 3, 0, 1)>(0, t);
        t += dpp<DPP_ROW_HALF_MIRROR>(0, t);
        total = t + dpp<DPP_ROW_MIRROR>(0, t);
    } else {
        // rows first (as DP == 16), then carry row totals across rows with shuffles
        incl += dpp<DPP_ROW_SHR(1)>(0, incl);
        incl += dpp<DPP_ROW_SHR(2)>(0, incl);
        incl += dpp<DPP_ROW_SHR(4)>(0, incl);
        incl += dpp<DPP_ROW_SHR(8)>(0, incl);
        const int row = lane_d >> 4;
        const uint32_t row_tot = __shfl(incl, (lane_d | 15), DP);          // lane 15 of my row
        uint32_t carry = 0, all = 0;
#pragma unroll
        for (int r = 0; r < DP / 16; r++) {
            const uint32_t tr = __shfl(row_tot, r * 16, DP);
            if (r < row) carry += tr;
            all += tr;
        }
        incl += carry;
        total = all;
    }
    return incl - x;
}

}  // namespace sprintz

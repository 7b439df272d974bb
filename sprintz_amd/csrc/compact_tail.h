// compact_tail.h -- the dense container built INSIDE the encode launch (sprintz_mi355x_compress_batch_dense).
//
// The two-launch write path (encode into worst-case slots, then scan the sizes and copy slot -> container) spends 0.17 of
// its 0.80 ms (headline batch) in the copy, after the encoder -- a compute-bound kernel -- has gone idle.  Here every
// workgroup finishes its own chunks' part of the container before it exits:
//   1. its chunks' (16-byte aligned) sizes are summed and scanned in LDS;
//   2. the workgroup's base in the container comes from a single-pass chained scan over workgroups (decoupled look-back:
//      one 64-bit word per workgroup, flag in the top two bits -- 1 "my own total", 2 "everything up to and including me" --
//      published and polled with relaxed device-scope atomics; nothing else crosses workgroups, so no fence is needed);
//   3. each wavefront copies its chunks slot -> container, 16 bytes a lane, while other workgroups still encode: the copy's
//      memory time hides under their arithmetic, and the slots are read back while they are still in this XCD's L2.
// Workgroups take their number from an atomic ticket, not from blockIdx: the look-back waits on LOWER numbers only, and a
// ticket holder is by construction already running.
// offsets[] and the container come out exactly as sprintz_mi355x_compact(align = 16) writes them.
#pragma once

#include "sprintz_device.h"

#ifndef SPRINTZ_DENSE_NT
#define SPRINTZ_DENSE_NT 0      // 1: the container is written with non-temporal stores
#endif

namespace sprintz {

struct DenseArgs {
    uint8_t* dense;             // container (null: the kernel writes slots only, the caller compacts)
    uint64_t* offsets;          // [nchunks + 1]
    uint64_t* wg_state;         // [grid + 1], zeroed before the launch; wg_state[grid] is the ticket counter
    uint32_t grid;
};

// this workgroup's number: the ticket when the dense tail is on (see above), blockIdx.x otherwise.  ALL threads call it.
__device__ __forceinline__ uint32_t workgroup_number(const DenseArgs& da)
{
    if (!da.dense) return blockIdx.x;
    __shared__ uint32_t s_ticket;
    if (threadIdx.x == 0)
        s_ticket = (uint32_t)__hip_atomic_fetch_add(&da.wg_state[da.grid], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    return s_ticket;
}

constexpr uint64_t kFlagAggregate = 1ull << 62, kFlagInclusive = 2ull << 62, kValueMask = (1ull << 62) - 1;

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, off, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), off, 64);
        v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}

// Called by EVERY thread of the workgroup once its chunk is encoded into its slot (threads whose chunk does not exist
// included).  wg: workgroup_number(); chunks of this workgroup are wg * CW + g, g = threadIdx.x >> log2_lanes;
// size: the chunk's stream bytes (group-uniform; 0 for a chunk that does not exist); lds: >= 16 bytes per group at
// lds + g * lds_stride that the kernel no longer needs.  CW = kThreads >> log2_lanes must be <= 64.
__device__ __forceinline__ void dense_tail(const DenseArgs& da, uint32_t wg, uint64_t nchunks, uint32_t log2_lanes, uint32_t size,
                                           const uint8_t* slots, uint64_t slot_stride, uint8_t* lds, uint32_t lds_stride)
{
    const uint32_t CW = (uint32_t)kThreads >> log2_lanes;
    const uint32_t g = threadIdx.x >> log2_lanes, lane_in_group = threadIdx.x & ((1u << log2_lanes) - 1u);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t c0 = (uint64_t)wg * CW;
    auto slot_of = [&](uint32_t grp) { return (uint64_t*)(lds + (size_t)grp * lds_stride); };

    __syncthreads();                                       // every group is done with its LDS; every slot store is out
    const uint32_t asize = (size + 15u) & ~15u;
    if (lane_in_group == 0) slot_of(g)[0] = asize;
    __syncthreads();
    if (wave == 0) {
        // exclusive scan of the CW sizes (lane g holds group g's), total in lane CW - 1
        uint64_t mine = lane < CW ? slot_of(lane)[0] : 0ull, incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)incl, off, 64), hi = (uint32_t)__shfl_up((int)(uint32_t)(incl >> 32), off, 64);
            if ((int)lane >= off) incl += ((uint64_t)hi << 32) | lo;
        }
        const uint64_t total = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(incl >> 32), 63, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)incl, 63, 64);
        // ---- chained scan over workgroups
        if (lane == 0)
            __hip_atomic_store(&da.wg_state[wg], (wg == 0 ? kFlagInclusive : kFlagAggregate) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint64_t base = 0;
        int64_t j = (int64_t)wg - 1;
        while (j >= 0) {
            const int64_t idx = j - (int64_t)lane;
            uint64_t v = kFlagInclusive;                   // before workgroup 0: an inclusive prefix of 0
            if (idx >= 0) {
                v = __hip_atomic_load(&da.wg_state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while ((v >> 62) == 0) {                   // not published yet: its holder is running (tickets), so this ends
                    __builtin_amdgcn_s_sleep(1);
                    v = __hip_atomic_load(&da.wg_state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            const uint64_t incl_lanes = __ballot((v >> 62) == 2);
            if (incl_lanes) {                              // the nearest predecessor that already knows its inclusive prefix
                const uint32_t first = (uint32_t)__builtin_ctzll(incl_lanes);
                base += wave_sum_u64(lane <= first ? (v & kValueMask) : 0ull);
                break;
            }
            base += wave_sum_u64(v & kValueMask);
            j -= 64;
        }
        if (lane == 0 && wg != 0)
            __hip_atomic_store(&da.wg_state[wg], kFlagInclusive | (base + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane < CW) slot_of(lane)[1] = base + (incl - mine);          // where group `lane`'s stream goes
    }
    __syncthreads();
    const uint64_t off = slot_of(g)[1];
    const uint64_t chunk = c0 + g;
    if (lane_in_group == 0 && chunk < nchunks) {
        da.offsets[chunk] = off;
        if (chunk == nchunks - 1) da.offsets[nchunks] = off + asize;
    }
    // ---- the copy: a wavefront takes its own groups' chunks one after the other, 64 lanes on each
    const uint32_t groups_per_wave = log2_lanes >= 6 ? 1u : 64u >> log2_lanes;
    const uint32_t g_first = log2_lanes >= 6 ? g : wave * groups_per_wave;
    for (uint32_t k = 0; k < groups_per_wave; k++) {
        const uint32_t gg = g_first + k;
        const uint64_t cc = c0 + gg;
        if (cc >= nchunks) break;
        const uint32_t n = (uint32_t)slot_of(gg)[0];
        const uint64_t o = slot_of(gg)[1];
        const uint32_t l = log2_lanes >= 6 ? threadIdx.x & ((1u << log2_lanes) - 1u) : lane;
        copy_verbatim<SPRINTZ_DENSE_NT != 0>(slots + cc * slot_stride, da.dense + o, n, l, log2_lanes >= 6 ? 1u << log2_lanes : 64u);
    }
}

}  // namespace sprintz

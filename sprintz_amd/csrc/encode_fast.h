// encode_fast.h -- batched encoder for the general (row-major payload) layout
// with one column per lane (D <= 64, 16-byte aligned blocks): same stream bytes
// as encode_kernel.h (sprintz_xff_rle.cpp:61-555, sprintz_delta_rle.cpp:55-404),
// cheaper per sample:
//   * INPUT.  An 8 x D block of the raw input is contiguous (8*D*ESZ bytes): the
//     group loads it as one 16-byte piece per lane, one block AHEAD of the one
//     being analysed, transposes it through LDS (ds_write_b128 -> 8 x
//     ds_read_u16/u8) -- instead of 8 scattered 2-byte global loads per lane;
//   * the per-block nbits scan runs on DPP (group_ops.h);
//   * FIRE arithmetic pinned to v_mad_i32_i24 / v_med3_i32 (see decode_fast.h);
//   * OUTPUT: fields OR-ed with ds_or_b32 into a zeroed, LINEAR per-group LDS window
//     (plain offsets, one 64-bit shift per field), whole 16-byte pieces flushed to HBM
//     per stream group.
#pragma once

#include "compact_tail.h"
#include "decode_fast.h"
#include "encode_kernel.h"

namespace sprintz {

#ifndef SPRINTZ_ENC_PAIR_MERGE
#define SPRINTZ_ENC_PAIR_MERGE 1
#endif
#ifndef SPRINTZ_ENC_DRAIN_ALIGN
#define SPRINTZ_ENC_DRAIN_ALIGN 128
#endif

// CM: column-major source (EncodeArgs::col_stride): a lane's 8 samples of a block are
// contiguous in ITS column -- one 16-byte (8-byte at W == 8) load per lane, no LDS transpose.
template <int W, bool FIRE, int DP, bool EXACT, bool CM>
__device__ __forceinline__ uint32_t encode_fast_body(const EncodeArgs& a, uint32_t wg_number)
{
    using U = typename Elem<W>::U;
    constexpr int HB = Elem<W>::HB;
    constexpr int ESZ = W / 8;
    constexpr int LOG2DP = DP == 4 ? 2 : DP == 8 ? 3 : DP == 16 ? 4 : DP == 32 ? 5 : 6;
    constexpr bool TAIL_LE = FIRE;               // "<=" at sprintz_xff_rle.cpp:362, "<" at sprintz_delta_rle.cpp:226
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int D = EXACT ? DP : a.D;
    const uint64_t gtid = (uint64_t)wg_number * kThreads + threadIdx.x;
    const uint64_t chunk = gtid >> LOG2DP;
    const int lane_d = (int)(threadIdx.x & (uint32_t)(DP - 1));
    if (chunk >= a.nchunks) return 0;

    const uint64_t first = chunk * (uint64_t)a.chunk_len;
    const uint32_t n = (uint32_t)((a.total_len - first < a.chunk_len) ? (a.total_len - first) : a.chunk_len);
    const U* const sc = (const U*)a.src + first;
    uint8_t* const gdst = a.slots + chunk * a.slot_stride;
    const bool col_ok = EXACT ? true : lane_d < D;

    // LDS carve per group: [output window cap | input block staging].  The window is a
    // LINEAR, zero-initialised image of the stream from `gpos` (16-byte aligned) on:
    // bit fields are OR-ed into it at plain offsets (no ring arithmetic); whenever a
    // stream group starts, its whole 16-byte pieces go to HBM and the < 16-byte rest
    // moves to the front.  cap >= 16 + one group + the close-out bytes.
    const uint32_t cap = a.cap;
    uint8_t* const win = smem + (size_t)(threadIdx.x >> LOG2DP) * a.lds_group_stride;
    uint8_t* const stage = win + cap;
    const uint32_t win_a = lds_addr(win);

    for (uint32_t u = (uint32_t)lane_d; u < (cap >> 4); u += DP) ((uint4*)win)[u] = make_uint4(0, 0, 0, 0);
    wave_lds_sync();

    uint32_t wl = a.norle == 1 ? 6u : ((a.norle == 2 || a.write_size) ? 8u : 0u);   // write position inside the window (bytes)
    uint32_t gpos = 0;                        // stream offset of the window start (multiple of 16)

    // whole 16-byte pieces below `upto` (window offset, multiple of 16) -> HBM; re-zero; slide
    auto drain = [&](uint32_t upto) {
        wave_lds_sync();
        for (uint32_t u = (uint32_t)lane_d * 16u; u < upto; u += DP * 16) {
            uint4* r = (uint4*)(win + u);
#ifndef SPRINTZ_ABL_ENC_NO_STORE
            *(uint4*)(gdst + gpos + u) = *r;
#endif
            *r = make_uint4(0, 0, 0, 0);
        }
        wave_lds_sync();
        if (upto != 0 && upto < cap) {                     // what stays (less than the flush granule, so source and front do not overlap) moves to the front
            const uint32_t rest = (wl - upto + 15u) & ~15u;
            for (uint32_t u = (uint32_t)lane_d * 16u; u < rest && upto + u < cap; u += DP * 16) {
                const uint4 v = *(uint4*)(win + upto + u);
                *(uint4*)(win + upto + u) = make_uint4(0, 0, 0, 0);
                *(uint4*)(win + u) = v;
            }
        }
        gpos += upto;
        wl -= upto;
        wave_lds_sync();
    };
    // OR the low nb (<= 16) bits of v at window bit position bp: always two dwords (the
    // second one is an OR with 0 when the field does not straddle)
    auto or_bits = [&](uint32_t bp, uint32_t v, uint32_t nb) {
        if (nb == 0) return;
        const uint64_t x = (uint64_t)v << (bp & 31u);
        __attribute__((address_space(3))) uint32_t* q =
            (__attribute__((address_space(3))) uint32_t*)(uintptr_t)(win_a + ((bp >> 3) & ~3u));
        __hip_atomic_fetch_or(q, (uint32_t)x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        if ((uint32_t)(x >> 32)) __hip_atomic_fetch_or(q + 1, (uint32_t)(x >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    };
    auto put_run = [&](uint32_t run) {          // sprintz_xff_rle.cpp:377-384
        if (lane_d == 0) {
            win[wl] = (uint8_t)((run & 0x7fu) | (run > 0x7fu ? 0x80u : 0u));
            if (run > 0x7fu) win[wl + 1] = (uint8_t)(run >> 7);
        }
        wl += run > 0x7fu ? 2u : 1u;
    };

    const uint32_t hdr_bytes = (2u * (uint32_t)D * HB + 7u) >> 3;
    const uint32_t blk = 8u * (uint32_t)D;
    const uint32_t blk_bytes = blk * ESZ;
    const uint32_t lane16 = (uint32_t)lane_d * 16u;
    const int64_t limit = (int64_t)n - 2 * (int64_t)blk;    // last_full_group_start (:158)
    int64_t pos_in = 0;
    uint32_t ngroups = 0, run = 0, hdr_pos = 0;
    int slot = 0;
    uint32_t pv = 0;
    int pd = 0, ctr = 0;

    auto start_group = [&]() {
        ngroups++;
        // whole 128-byte LINES leave, never 16-byte pieces of one: with 16-byte granularity a group's ~96 bytes ended mid-line and
        // the line was written twice, half each time (WRITE_SIZE 650 MB for 472 MB of streams; half-line stores are what the memory
        // system likes least -- decode_fast's ablation) -- headline compress 0.768 -> 0.722 ms in the same-box A/B
        drain(wl & ~(uint32_t)(SPRINTZ_ENC_DRAIN_ALIGN - 1));
        hdr_pos = wl;
        wl += hdr_bytes;
        slot = 0;
    };
    // this lane's 16-byte pieces of the 8 x D block starting at element `pos` (0 past the chunk)
    constexpr int PIECES = (8 * DP * ESZ + DP * 16 - 1) / (DP * 16);   // 1 (W=16) or 1 (W=8): block <= DP*16 bytes
    const U* const cmcol = CM ? (const U*)a.src + first / (uint64_t)D + (uint64_t)lane_d * a.col_stride : nullptr;
    auto load_block = [&](int64_t pos) -> uint4 {
        uint4 v = make_uint4(0, 0, 0, 0);
        if constexpr (CM) {
            if (col_ok && pos + (int64_t)blk <= (int64_t)n) {
                const U* p = cmcol + (uint32_t)pos / (uint32_t)D;          // 8 consecutive rows of this lane's column
                if constexpr (W == 16) v = *(const uint4*)p;
                else { const uint2 t = *(const uint2*)p; v.x = t.x; v.y = t.y; }
            }
        } else {
            if (lane16 < blk_bytes && pos + (int64_t)blk <= (int64_t)n) v = *(const uint4*)((const uint8_t*)(sc + pos) + lane16);
        }
        return v;
    };
    static_assert(PIECES == 1, "one 16-byte piece per lane covers a block");
    // Column-major source: a lone 16-byte load per lane per block makes every request its own cache
    // line (see decode_fast.h).  So the blocks arrive four at a time: a QUAD of lanes loads one
    // column's 64 contiguous bytes (= 4 blocks; 32 bytes at 8 bits) one burst ahead, the pieces wait
    // in LDS as cst[buffer][column][slot] (slots rotated by column/4, as in the decoder), and every
    // lane picks up its own column's block from there.
    constexpr uint32_t PB = W == 16 ? 16u : 8u;
    uint8_t* const cst = stage;                            // 4 blocks x DP columns x PB bytes
    // (ONE buffer: a burst is parked at the block boundary where the last block of the burst before it has just been taken, in a
    //  wavefront-synchronous step -- the second buffer this used to keep only cost LDS, i.e. resident waves)
    auto cst_at = [&](uint32_t, uint32_t col, uint32_t b4) { return cst + (col * 4u + ((b4 + (col >> 2)) & 3u)) * PB; };
    uint4 burst[4];                                        // this lane's share of the burst in flight
    const uint32_t cs_elems = CM ? (uint32_t)a.col_stride : 0u;
    const U* const cm0 = CM ? (const U*)a.src + first / (uint64_t)D : nullptr;
    auto burst_load = [&](uint32_t k) {                    // blocks 4k .. 4k+3
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t pid = (uint32_t)j * DP + (uint32_t)lane_d, col = pid >> 2, part = pid & 3u;
            const int64_t pos = ((int64_t)k * 4 + part) * (int64_t)blk;
            burst[j] = make_uint4(0, 0, 0, 0);
            if (col < (uint32_t)D && pos + (int64_t)blk <= (int64_t)n) {
                const U* p = cm0 + (uint64_t)col * cs_elems + (uint32_t)pos / (uint32_t)D;
                if constexpr (W == 16) burst[j] = *(const uint4*)p;
                else { const uint2 t = *(const uint2*)p; burst[j].x = t.x; burst[j].y = t.y; }
            }
        }
    };
    auto burst_park = [&](uint32_t k) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t pid = (uint32_t)j * DP + (uint32_t)lane_d, col = pid >> 2, part = pid & 3u;
            if (col < (uint32_t)D) {
                if constexpr (W == 16) *(uint4*)cst_at(k & 1u, col, part) = burst[j];
                else *(uint2*)cst_at(k & 1u, col, part) = make_uint2(burst[j].x, burst[j].y);
            }
        }
    };
    uint32_t bno = 0;                                      // blocks taken so far (= pos_in / blk)

    bool active = n >= 128u && limit >= 0;      // :116 and the loop guard :160
    uint4 nxt = make_uint4(0, 0, 0, 0);
    if (active) {
        start_group();
        if constexpr (CM) {
            burst_load(0);
            burst_park(0);
            burst_load(1);
            wave_lds_sync();
        } else {
            nxt = load_block(0);
        }
    }
    uint8_t* const stage_col = stage + lane_d * ESZ;
    const uint32_t row_stride = (uint32_t)D * ESZ;

    while (active) {
        // ---- the block at pos_in is in `nxt`; transpose it through LDS, request the next one
        uint32_t x[8];
        if constexpr (CM) {
            const uint32_t k = bno >> 2;
            if ((bno & 3u) == 0 && bno != 0) {             // a new burst starts: park it, request the one after
                wave_lds_sync();
                burst_park(k);
                burst_load(k + 1);
                wave_lds_sync();
            }
            uint4 cur = make_uint4(0, 0, 0, 0);
            if (col_ok) {
                if constexpr (W == 16) cur = *(const uint4*)cst_at(k & 1u, (uint32_t)lane_d, bno & 3u);
                else { const uint2 t = *(const uint2*)cst_at(k & 1u, (uint32_t)lane_d, bno & 3u); cur.x = t.x; cur.y = t.y; }
            }
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if constexpr (W == 16) {
                    const uint32_t d = (i >> 1) == 0 ? cur.x : (i >> 1) == 1 ? cur.y : (i >> 1) == 2 ? cur.z : cur.w;
                    x[i] = (i & 1) ? d >> 16 : d & 0xffffu;
                } else {
                    const uint32_t d = (i >> 2) == 0 ? cur.x : cur.y;
                    x[i] = (d >> (8 * (i & 3))) & 0xffu;
                }
            }
        } else {
            if (lane16 < blk_bytes) *(uint4*)(stage + lane16) = nxt;
            wave_lds_sync();
            nxt = load_block(pos_in + blk);
#pragma unroll
            for (int i = 0; i < 8; i++) x[i] = col_ok ? (uint32_t)*(const U*)(stage_col + i * row_stride) : 0u;
            wave_lds_sync();
        }

        // ---- forecast + zigzag + OR-mask (:197-298)
        uint32_t z[8];
        const int coef = FIRE ? fire_coef<W, false>(ctr) : 0;
        int grad = 0;
        uint32_t mask = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int delta = sext<W>((int)(x[i] - pv));
            int err;
            if constexpr (FIRE) {
                const int pred = __builtin_amdgcn_sbfe(mad24(pd, coef, 0), W, W);
                err = sext<W>(delta - pred);
                if (i & 1) grad = mad24(sign_of(err), pd, grad);
            } else {
                err = delta;
            }
            const uint32_t zz = zigzag<W>(err);
            mask |= zz;
            z[i] = zz;
            pv = x[i];
            pd = delta;
        }
        if constexpr (FIRE) ctr = wrap_counter<W>(ctr + __builtin_amdgcn_sbfe(grad, 2, W - 2));
        const uint32_t nb = col_ok ? nbits_of<W, false>(mask) : 0u;
        uint32_t total;
        const uint32_t excl = group_scan<DP>(nb, lane_d, total);

        // ---- RLE state machine (:350-456); group-uniform
        for (;;) {
            if (total == 0 && run < 0x7fffu && !a.norle) {     // (run-less codecs: an all-zero block is two empty header fields)
                run++;
                pos_in += blk;
                bno++;
                const bool more = TAIL_LE ? (pos_in <= limit) : (pos_in < limit);
                if (more) break;
                slot++;
                put_run(run);
                wl += (uint32_t)(2 - slot);
                run = 0;
                active = false;
                break;
            }
            if (run > 0) {
                slot++;
                put_run(run);
                run = 0;
                if (slot == 2) start_group();            // :430-450
                continue;
            }
            {   // header fields: HB bits per column, eight adjacent lanes = one 24/32-bit word, merged the same way
                uint32_t f = col_ok ? (nb == (uint32_t)W ? (uint32_t)(W - 1) : nb) : 0u;   // :296
                const uint32_t hbp = hdr_pos * 8u + (uint32_t)(slot * D + lane_d) * HB;
                if (SPRINTZ_ENC_PAIR_MERGE && DP >= 8) {
                    f |= dpp<DPP_ROW_SHL(1)>(0, f) << HB;
                    f |= dpp<DPP_ROW_SHL(2)>(0, f) << (2 * HB);
                    f |= dpp<DPP_ROW_SHL(4)>(0, f) << (4 * HB);
                    if ((lane_d & 7) == 0) or_bits(hbp, f, 8 * HB);
                } else if (col_ok) {
                    or_bits(hbp, f, HB);
                }
            }
            const uint32_t row_bits = ((total + 7u) >> 3) << 3;
            uint32_t bp = wl * 8u + excl;
            // Adjacent columns are adjacent bit fields of the same row: eight lanes OR-ing into the
            // same one or two dwords serialise in the LDS (SQ_LDS_ADDR_CONFLICT was half of its busy
            // cycles).  So a lane pair merges its two fields (<= 32 bits) in registers and only the
            // even lane issues the OR: half the same-address collisions for 2 VALU per row (cfg2 compress
            // + compact 1.45 -> 1.65 TB/s; merging a whole quad into 64 bits costs more VALU than it
            // saves: 1.50).
            if (SPRINTZ_ENC_PAIR_MERGE) {
                const uint32_t nb_pair = nb + dpp<DPP_ROW_SHL(1)>(0, nb);
                const bool even = (lane_d & 1) == 0;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const uint32_t zn = dpp<DPP_ROW_SHL(1)>(0, z[i]);
                    const uint32_t zz = z[i] | (nb < 32u ? zn << nb : 0u);
                    if (even) or_bits(bp, zz, nb_pair);
                    bp += row_bits;
                }
            } else {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                or_bits(bp, z[i], nb);
                bp += row_bits;
            }
            }
            wl += row_bits;                              // 8 rows * row_bytes
            pos_in += blk;
            bno++;
            slot++;
            if (slot == 2) {
                if (pos_in <= limit) start_group();
                else active = false;
            }
            break;
        }
    }

    // ---- verbatim tail through the window (:553)
    const uint32_t remaining = (uint32_t)((int64_t)n - pos_in);
    {
        const uint8_t* tp = (const uint8_t*)(sc + pos_in);
        uint32_t left = remaining * ESZ;
        while (left > 0) {
            drain(wl & ~15u);
            const uint32_t room = cap - 16u - wl;
            const uint32_t m = left < room ? left : room;
            if constexpr (CM) {
                const uint32_t done = remaining * ESZ - left;
                const U* const c0 = (const U*)a.src + first / (uint64_t)D;
                for (uint32_t j = (uint32_t)lane_d; j < m; j += DP) {
                    const uint32_t tb = done + j, e = (uint32_t)pos_in + tb / ESZ;
                    const uint32_t xv = (uint32_t)c0[(uint64_t)(e % (uint32_t)D) * a.col_stride + e / (uint32_t)D];
                    win[wl + j] = (uint8_t)(xv >> (8u * (tb % ESZ)));
                }
            } else {
                for (uint32_t j = (uint32_t)lane_d; j < m; j += DP) win[wl + j] = tp[j];
            }
            wl += m;
            tp += m;
            left -= m;
        }
    }
    const uint32_t total_bytes = gpos + wl;
    drain((wl + 15u) & ~15u);

    if (lane_d == 0) {                                   // format.h:36-45; lane 0 also flushed unit 0
        if (a.norle == 2) {                                  // u64 len with ndims in its bytes 6..7 (sprintz_xff.cpp:58-63)
            ((uint32_t*)gdst)[0] = n;
            ((uint32_t*)gdst)[1] = (uint32_t)D << 16;
        } else if (a.norle) {                                // {u32 len; u16 ndims} (format.h:65-72); bytes 6.. are stream
            ((uint32_t*)gdst)[0] = n;
            ((uint16_t*)gdst)[2] = (uint16_t)D;
        } else if (a.write_size) {
            ((uint32_t*)gdst)[0] = ngroups;
            ((uint32_t*)gdst)[1] = (remaining & 0xffffu) | ((uint32_t)D << 16);
        }
        a.sizes[chunk] = total_bytes;
        if (a.rets) a.rets[chunk] = (int64_t)(total_bytes / ESZ);
    }
    return total_bytes;
}

template <int W, bool FIRE, int DP, bool EXACT, bool CM = false>
__global__ void __launch_bounds__(kThreads) encode_fast_kernel(EncodeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr uint32_t LOG2DP = DP == 4 ? 2 : DP == 8 ? 3 : DP == 16 ? 4 : DP == 32 ? 5 : 6;
    const uint32_t wg = workgroup_number(a.dn);
    const uint32_t size = encode_fast_body<W, FIRE, DP, EXACT, CM>(a, wg);
    // the container, built before the workgroup leaves (compact_tail.h); without it the caller compacts the slots
    if (a.dn.dense) dense_tail(a.dn, wg, a.nchunks, LOG2DP, size, a.slots, a.slot_stride, smem, a.lds_group_stride);
}

}  // namespace sprintz

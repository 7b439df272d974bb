// encode_wide.h -- encode_fast.h's row-major path with TWO columns per lane, for streams of
// 65 .. 128 columns (BASELINE config 3 has 80; the paper's MSRC-12 has 80 at either width): one 64-lane
// group per chunk, lane l owns columns 2l and 2l + 1.  Same stream bytes as encode_kernel.h (sprintz_xff_rle.cpp:61-555,
// sprintz_delta_rle.cpp:55-404).  An 8 x D block is at most 1 024 bytes (2 048 at 16 bits: two pieces
// per lane), so the input side is unchanged (16-byte pieces a block ahead, transposed through LDS); on the output side a
// lane's two fields are adjacent in the row and are merged in registers for nothing, and a lane pair
// merges its four (<= 32 bits at 8 bits; at 16 bits a lane's own two already fill a dword) with one DPP
// move, so a quarter (half) of the columns' worth of LDS ORs is issued and same-address collisions
// shrink accordingly.
#pragma once

#include "encode_fast.h"
#include "decode_fast.h"      // mad_i16_hi

#ifndef SPRINTZ_ENC_HIGH_HALF
#define SPRINTZ_ENC_HIGH_HALF 0      // round 5: built, bit-exact, measured 1 % SLOWER on the headline shape (0.684 / 0.671 / 0.675 against 0.671 / 0.667 / 0.659 ms, tools/ab_list.sh): off
#endif

namespace sprintz {

// SPLIT (8 bits, 65 .. 80 columns -- 40 of the 64 lanes carry columns at 80): 32 lanes a chunk, two chunks a wavefront; lane l
// owns the PAIR (2l, 2l + 1) -- merged with its neighbour's exactly as above -- and the SINGLE column 64 + l, whose fields a
// quad of lanes merges (<= 32 bits) before one of them ORs; one scan carries the pairs' bits in its low half and the singles' in
// its high half.  The decoder's counterpart is decode_fast.h's SPLIT mapping.
// DPT: lanes a chunk when that is not 64 (two columns per lane for NARROW streams too: 4 lanes for 5 .. 8 columns, 16 chunks a wavefront).
// CM: column-major source (EncodeArgs::col_stride), as in encode_fast.h: a quad of lanes loads one column's 64 contiguous bytes
// (four blocks; 32 bytes at 8 bits) one burst ahead, the pieces wait in LDS, and a lane takes its two columns' blocks from there.
template <int W, bool FIRE, bool EXACT, bool SPLIT, int DPT, bool CM = false>
__device__ __forceinline__ uint32_t encode_wide_body(const EncodeArgs& a, uint32_t wg_number)
{
    static_assert(!(CM && SPLIT), "column-major sources: two columns per lane only");
    constexpr int DP = SPLIT ? 32 : DPT, CPL = SPLIT ? 3 : 2;
    constexpr int LOG2DP = DP == 4 ? 2 : DP == 8 ? 3 : DP == 16 ? 4 : DP == 32 ? 5 : 6;
    static_assert(DP == 4 || DP == 8 || DP == 16 || DP == 32 || DP == 64, "lanes a chunk");
    static_assert(!SPLIT || (W == 8 && !EXACT), "the split mapping is built for 8-bit streams of 65 .. 80 columns");
    using U = typename Elem<W>::U;
    constexpr int HB = Elem<W>::HB;
    constexpr int ESZ = W / 8;
    constexpr int PIECES = SPLIT ? 2 : ESZ;      // 16-byte pieces of a block per lane: 8 * 128 * ESZ bytes / (64 * 16); split: 8 * 80 / (32 * 16)
    constexpr bool TAIL_LE = FIRE;               // "<=" at sprintz_xff_rle.cpp:362, "<" at sprintz_delta_rle.cpp:226
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int D = EXACT ? DP * CPL : a.D;
    const uint64_t gtid = (uint64_t)wg_number * kThreads + threadIdx.x;
    const uint64_t chunk = gtid >> LOG2DP;
    const int lane_d = (int)(threadIdx.x & (uint32_t)(DP - 1));
    if (chunk >= a.nchunks) return 0;

    const uint64_t first = chunk * (uint64_t)a.chunk_len;
    const uint32_t n = (uint32_t)((a.total_len - first < a.chunk_len) ? (a.total_len - first) : a.chunk_len);
    const U* const sc = (const U*)a.src + first;
    uint8_t* const gdst = a.slots + chunk * a.slot_stride;
    const int col0 = SPLIT ? 2 * lane_d : lane_d * CPL;    // first column of the lane's pair
    int genk[CPL];
    bool col_ok[CPL];
#pragma unroll
    for (int k = 0; k < CPL; k++) {
        genk[k] = (SPLIT && k == 2) ? 64 + lane_d : col0 + k;
        col_ok[k] = EXACT ? true : genk[k] < D;
    }

    // LDS per group: [linear, zero-initialised output window cap | input block staging] (encode_fast.h)
    const uint32_t cap = a.cap;
    uint8_t* const win = smem + (size_t)(threadIdx.x >> LOG2DP) * a.lds_group_stride;
    uint8_t* const stage = win + cap;
    const uint32_t win_a = lds_addr(win);
    for (uint32_t u = (uint32_t)lane_d; u < (cap >> 4); u += DP) ((uint4*)win)[u] = make_uint4(0, 0, 0, 0);
    wave_lds_sync();

    uint32_t wl = a.norle == 1 ? 6u : ((a.norle == 2 || a.write_size) ? 8u : 0u);
    uint32_t gpos = 0;
    auto drain = [&](uint32_t upto) {            // whole 16-byte pieces below `upto` -> HBM; re-zero; slide
        wave_lds_sync();
        for (uint32_t u = (uint32_t)lane_d * 16u; u < upto; u += DP * 16) {
            uint4* r = (uint4*)(win + u);
            *(uint4*)(gdst + gpos + u) = *r;
            *r = make_uint4(0, 0, 0, 0);
        }
        wave_lds_sync();
        if (upto != 0 && upto < cap) {                     // what stays (less than the flush granule) moves to the front
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
    auto or_bits = [&](uint32_t bp, uint32_t v, uint32_t nb) {      // low nb (<= 32) bits of v at window bit bp
#ifdef SPRINTZ_ENC_ABL_NO_OR                       // ablation builds (tools/build_variant.sh): what the ORs cost / what their same-dword collisions cost
        return;
#endif
#ifdef SPRINTZ_ENC_ABL_SPREAD_OR
        bp += (uint32_t)lane_d * 64u;
#endif
#ifndef SPRINTZ_ENC_OR_NOZERO
        if (nb == 0) return;
#endif
        const uint64_t x = (uint64_t)v << (bp & 31u);
        __attribute__((address_space(3))) uint32_t* q =
            (__attribute__((address_space(3))) uint32_t*)(uintptr_t)(win_a + ((bp >> 3) & ~3u));
        __hip_atomic_fetch_or(q, (uint32_t)x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#ifdef SPRINTZ_ENC_OR_BOTH
        __hip_atomic_fetch_or(q + 1, (uint32_t)(x >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#else
        if ((uint32_t)(x >> 32)) __hip_atomic_fetch_or(q + 1, (uint32_t)(x >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#endif
    };
    auto put_run = [&](uint32_t run) {           // sprintz_xff_rle.cpp:377-384
        if (lane_d == 0) {
            win[wl] = (uint8_t)((run & 0x7fu) | (run > 0x7fu ? 0x80u : 0u));
            if (run > 0x7fu) win[wl + 1] = (uint8_t)(run >> 7);
        }
        wl += run > 0x7fu ? 2u : 1u;
    };

    const uint32_t hdr_bytes = (2u * (uint32_t)D * HB + 7u) >> 3;
    const uint32_t blk = 8u * (uint32_t)D;       // elements per block
    const uint32_t blk_bytes = blk * ESZ;
    const uint32_t lane16 = (uint32_t)lane_d * 16u;
    const int64_t limit = (int64_t)n - 2 * (int64_t)blk;
    int64_t pos_in = 0;
    uint32_t ngroups = 0, run = 0, hdr_pos = 0;
    int slot = 0;
    uint32_t pv[CPL];
    int pd[CPL], ctr[CPL];
#pragma unroll
    for (int k = 0; k < CPL; k++) { pv[k] = 0; pd[k] = 0; ctr[k] = 0; }
    auto start_group = [&]() {
        ngroups++;
        drain(wl & ~(uint32_t)(SPRINTZ_ENC_DRAIN_ALIGN - 1));   // whole 128-byte lines (encode_fast.h)
        hdr_pos = wl;
        wl += hdr_bytes;
        slot = 0;
    };
    // ---- column-major source: bursts of four blocks (encode_fast.h; one staging buffer)
    constexpr uint32_t PB = W == 16 ? 16u : 8u;           // bytes of one block of one column
    uint8_t* const cst = stage;                            // DP * CPL columns x 4 blocks x PB bytes
    auto cst_at = [&](uint32_t col, uint32_t b4) { return cst + (col * 4u + ((b4 + (col >> 2)) & 3u)) * PB; };
    uint4 burst[CM ? 4 * CPL : 1];
    const uint32_t cs_elems = CM ? (uint32_t)a.col_stride : 0u;
    const U* const cm0 = CM ? (const U*)a.src + first / (uint64_t)D : nullptr;      // column 0 at this chunk's first row
    auto burst_load = [&](uint32_t k) {                    // blocks 4k .. 4k+3 of every column
        if constexpr (CM) {
#pragma unroll
            for (int j = 0; j < 4 * CPL; j++) {
                const uint32_t pid = (uint32_t)j * DP + (uint32_t)lane_d, col = pid >> 2, part = pid & 3u;
                const int64_t pos = ((int64_t)k * 4 + part) * (int64_t)(8u * (uint32_t)D);
                burst[j] = make_uint4(0, 0, 0, 0);
                if (col < (uint32_t)D && pos + (int64_t)(8u * (uint32_t)D) <= (int64_t)n) {
                    const U* p = cm0 + (uint64_t)col * cs_elems + (uint32_t)pos / (uint32_t)D;
                    if constexpr (W == 16) burst[j] = *(const uint4*)p;
                    else { const uint2 t = *(const uint2*)p; burst[j].x = t.x; burst[j].y = t.y; }
                }
            }
        }
    };
    auto burst_park = [&]() {
        if constexpr (CM) {
#pragma unroll
            for (int j = 0; j < 4 * CPL; j++) {
                const uint32_t pid = (uint32_t)j * DP + (uint32_t)lane_d, col = pid >> 2, part = pid & 3u;
                if (col < (uint32_t)D) {
                    if constexpr (W == 16) *(uint4*)cst_at(col, part) = burst[j];
                    else *(uint2*)cst_at(col, part) = make_uint2(burst[j].x, burst[j].y);
                }
            }
        }
    };
    uint32_t bno = 0;                                      // blocks taken so far (= pos_in / blk)

    uint4 nxt[PIECES];
    auto load_block = [&](int64_t pos) {
#pragma unroll
        for (int q = 0; q < PIECES; q++) {
            const uint32_t u = lane16 + (uint32_t)q * DP * 16u;
            nxt[q] = make_uint4(0, 0, 0, 0);
            if (u < blk_bytes && pos + (int64_t)blk <= (int64_t)n) nxt[q] = *(const uint4*)((const uint8_t*)(sc + pos) + u);
        }
    };

    bool active = n >= 128u && limit >= 0;
#pragma unroll
    for (int q = 0; q < PIECES; q++) nxt[q] = make_uint4(0, 0, 0, 0);
    if (active) {
        start_group();
        if constexpr (CM) {
            burst_load(0);
            burst_park();
            burst_load(1);
            wave_lds_sync();
        } else {
            load_block(0);
        }
    }
    const uint32_t row_stride = (uint32_t)D;     // elements

    while (active) {
        // ---- the block at pos_in is in `nxt`: transpose it through LDS, request the next one
        uint4 cmcur[CPL];
        if constexpr (CM) {
            if ((bno & 3u) == 0 && bno != 0) {             // a new burst starts: park it (the one before it has been taken), request the one after
                wave_lds_sync();
                burst_park();
                burst_load((bno >> 2) + 1);
                wave_lds_sync();
            }
#pragma unroll
            for (int k = 0; k < CPL; k++) {
                cmcur[k] = make_uint4(0, 0, 0, 0);
                if (col_ok[k]) {
                    if constexpr (W == 16) cmcur[k] = *(const uint4*)cst_at((uint32_t)genk[k], bno & 3u);
                    else { const uint2 t = *(const uint2*)cst_at((uint32_t)genk[k], bno & 3u); cmcur[k].x = t.x; cmcur[k].y = t.y; }
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < PIECES; q++) {
                const uint32_t u = lane16 + (uint32_t)q * DP * 16u;
                if (u < blk_bytes) *(uint4*)(stage + u) = nxt[q];
            }
            wave_lds_sync();
            load_block(pos_in + blk);
        }
        uint32_t z[CPL][8], nb[CPL], lane_bits = 0;
#if SPRINTZ_ENC_HIGH_HALF
        // Round 5, 16-bit pairs (EXACT shapes: rows are whole dwords): the forecast in the HIGH HALF of the registers.  A row's two samples are one
        // LDS dword; a sample x lives as X = x << 16, so the 16-bit wrap of every difference is the 32-bit subtraction's own (no sign
        // extension: three v_bfe_i32 a sample less), the prediction's (pd * coef) >> 16 is one v_mad_i32_i16 that takes pd from PD's
        // high half (op_sel) and a mask, and sign(err) * pd rides the same instruction.  pv / pd hold X and the shifted delta here.
        constexpr bool kHigh = W == 16 && EXACT && !CM && !SPLIT;
#else
        constexpr bool kHigh = false;
#endif
        if constexpr (kHigh) {
            uint32_t dq[8];
#pragma unroll
            for (int i = 0; i < 8; i++) dq[i] = *(const uint32_t*)(stage + 4u * (uint32_t)lane_d + (uint32_t)i * row_stride * 2u);
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int coef = FIRE ? fire_coef<W, false>(ctr[k]) : 0;
                int grad = 0;
                uint32_t mask = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const uint32_t X = k == 0 ? dq[i] << 16 : dq[i] & 0xffff0000u;
                    const int DELTA = (int)(X - pv[k]);
                    int ERR;
                    if constexpr (FIRE) {
                        const int prod = mad_i16_hi(pd[k], coef, 0);                  // pd * coef, pd = the high half of the shifted delta
                        ERR = DELTA - (int)((uint32_t)prod & 0xffff0000u);
                        if (i & 1) grad = mad_i16_hi(pd[k], ERR > 0 ? 1 : (ERR < 0 ? -1 : 0), grad);
                    } else {
                        ERR = DELTA;
                    }
                    const uint32_t zz = (((uint32_t)ERR >> 15) ^ (uint32_t)(ERR >> 31)) & 0xffffu;
                    mask |= zz;
                    z[k][i] = zz;
                    pv[k] = X;
                    pd[k] = DELTA;
                }
                if constexpr (FIRE) ctr[k] = wrap_counter<W>(ctr[k] + __builtin_amdgcn_sbfe(grad, 2, W - 2));
                nb[k] = nbits_of<W, false>(mask);
                lane_bits += nb[k];
            }
        }
        uint32_t xp[8];                                    // 8 bits: the pair's two samples of a row come as ONE 16-bit LDS read
        if constexpr (W == 8 && !CM) {                     // (rows are an even number of bytes -- blocks are 16-byte multiples -- and the pair starts on an even column)
#pragma unroll
            for (int i = 0; i < 8; i++) xp[i] = col_ok[0] ? (uint32_t)*(const uint16_t*)(stage + (uint32_t)genk[0] + i * row_stride) : 0u;
        }
#pragma unroll
        for (int k = 0; k < (kHigh ? 0 : CPL); k++) {
            uint32_t x[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if constexpr (CM) {                        // the lane's own column block, eight samples in registers
                    if constexpr (W == 16) {
                        const uint32_t d = (i >> 1) == 0 ? cmcur[k].x : (i >> 1) == 1 ? cmcur[k].y : (i >> 1) == 2 ? cmcur[k].z : cmcur[k].w;
                        x[i] = (i & 1) ? d >> 16 : d & 0xffffu;
                    } else {
                        const uint32_t d = (i >> 2) == 0 ? cmcur[k].x : cmcur[k].y;
                        x[i] = (d >> (8 * (i & 3))) & 0xffu;
                    }
                } else if (W == 8 && k < 2) x[i] = col_ok[k] ? (k == 0 ? xp[i] & 0xffu : xp[i] >> 8) : 0u;
                else x[i] = col_ok[k] ? (uint32_t)((const U*)stage)[(uint32_t)genk[k] + i * row_stride] : 0u;
            }
            const int coef = FIRE ? fire_coef<W, false>(ctr[k]) : 0;
            int grad = 0;
            uint32_t mask = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int delta = sext<W>((int)(x[i] - pv[k]));
                int err;
                if constexpr (FIRE) {
                    const int pred = __builtin_amdgcn_sbfe(mad24(pd[k], coef, 0), W, W);
                    err = sext<W>(delta - pred);
                    if (i & 1) grad = mad24(sign_of(err), pd[k], grad);
                } else {
                    err = delta;
                }
                const uint32_t zz = zigzag<W>(err);
                mask |= zz;
                z[k][i] = zz;
                pv[k] = x[i];
                pd[k] = delta;
            }
            if constexpr (FIRE) ctr[k] = wrap_counter<W>(ctr[k] + __builtin_amdgcn_sbfe(grad, 2, W - 2));
            nb[k] = col_ok[k] ? nbits_of<W, false>(mask) : 0u;
            if (k < 2) lane_bits += nb[k];                 // the pair's bits (the single column of the split mapping is scanned beside them)
        }
        if constexpr (!CM) wave_lds_sync();
        uint32_t total, excl, excl_single = 0;
        if constexpr (SPLIT) {                             // columns 0 .. 63, lane by lane, then the singles: <= 512 and <= 256 bits
            uint32_t both;
            const uint32_t r = group_scan<DP>(lane_bits | (nb[2] << 16), lane_d, both);
            excl = r & 0xffffu;
            excl_single = (both & 0xffffu) + (r >> 16);
            total = (both & 0xffffu) + (both >> 16);
        } else {
            excl = group_scan<DP>(lane_bits, lane_d, total);
        }

        // ---- RLE state machine (:350-456); group-uniform
        for (;;) {
#ifndef SPRINTZ_ENC_ABL_NO_RLE                     // ablation build: no run handling at all (right only for data without zero blocks): what the state machine costs
            if (total == 0 && run < 0x7fffu && !a.norle) {
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
                if (slot == 2) start_group();
                continue;
            }
#endif
            {   // header fields: the lane's two, then four lanes' eight = one 24-bit word (:296)
                uint32_t f = 0;
#pragma unroll
                for (int k = 0; k < 2; k++) f |= (col_ok[k] ? (nb[k] == (uint32_t)W ? (uint32_t)(W - 1) : nb[k]) : 0u) << (k * HB);
                f |= dpp<DPP_ROW_SHL(1)>(0, f) << (2 * HB);
                f |= dpp<DPP_ROW_SHL(2)>(0, f) << (4 * HB);
                if ((lane_d & 3) == 0) or_bits(hdr_pos * 8u + (uint32_t)(slot * D + col0) * HB, f, 8 * HB);
                if constexpr (SPLIT) {                     // the singles: eight lanes' fields = one 24-bit word
                    uint32_t g = col_ok[2] ? (nb[2] == (uint32_t)W ? (uint32_t)(W - 1) : nb[2]) : 0u;
                    g |= dpp<DPP_ROW_SHL(1)>(0, g) << HB;
                    g |= dpp<DPP_ROW_SHL(2)>(0, g) << (2 * HB);
                    g |= dpp<DPP_ROW_SHL(4)>(0, g) << (4 * HB);
                    if ((lane_d & 7) == 0 && col_ok[2]) or_bits(hdr_pos * 8u + (uint32_t)(slot * D + genk[2]) * HB, g, 8 * HB);
                }
            }
            const uint32_t row_bits = ((total + 7u) >> 3) << 3;
            uint32_t bp = wl * 8u + excl;
            if constexpr (W == 8) {
                const uint32_t nb_pair = lane_bits + dpp<DPP_ROW_SHL(1)>(0, lane_bits);   // this lane's and the next one's: 4 fields <= 32 bits
                const bool even = (lane_d & 1) == 0;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const uint32_t mine = z[0][i] | (z[1][i] << nb[0]);              // <= 16 bits
                    const uint32_t theirs = dpp<DPP_ROW_SHL(1)>(0, mine);
                    if (even) or_bits(bp, mine | (theirs << lane_bits), nb_pair);
                    bp += row_bits;
                }
                if constexpr (SPLIT) {                     // a quad of lanes merges its four single-column fields (<= 32 bits)
                    const uint32_t c1 = nb[2], c2 = c1 + dpp<DPP_ROW_SHL(1)>(0, c1), c4 = c2 + dpp<DPP_ROW_SHL(2)>(0, c2);
                    const bool lead = (lane_d & 3) == 0;
                    uint32_t bs = wl * 8u + excl_single;
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const uint32_t m2 = z[2][i] | (dpp<DPP_ROW_SHL(1)>(0, z[2][i]) << c1);   // <= 16 bits
                        const uint32_t m4 = m2 | (c2 < 32u ? dpp<DPP_ROW_SHL(2)>(0, m2) << c2 : 0u);
                        if (lead) or_bits(bs, m4, c4);
                        bs += row_bits;
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    or_bits(bp, z[0][i] | (nb[0] < 32u ? z[1][i] << nb[0] : 0u), lane_bits);   // the lane's own two fields: <= 32 bits
                    bp += row_bits;
                }
            }
            wl += row_bits;
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
            if constexpr (CM) {                            // the tail continues the row-major order: element e sits in column e % D
                const uint32_t done = remaining * ESZ - left;
                for (uint32_t j = (uint32_t)lane_d; j < m; j += DP) {
                    const uint32_t tb = done + j, e = (uint32_t)pos_in + tb / ESZ;
                    const uint32_t xv = (uint32_t)cm0[(uint64_t)(e % (uint32_t)D) * a.col_stride + e / (uint32_t)D];
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
        if (a.norle == 2) {                              // u64 len with ndims in its bytes 6..7 (sprintz_xff.cpp:58-63)
            ((uint32_t*)gdst)[0] = n;
            ((uint32_t*)gdst)[1] = (uint32_t)D << 16;
        } else if (a.norle) {                            // {u32 len; u16 ndims} (format.h:65-72)
            ((uint32_t*)gdst)[0] = n;
            ((uint16_t*)gdst)[2] = (uint16_t)D;
        } else if (a.write_size) {
            ((uint32_t*)gdst)[0] = ngroups;
            ((uint32_t*)gdst)[1] = (remaining & 0xffffu) | ((uint32_t)D << 16);
        }
        a.sizes[chunk] = total_bytes;
        if (a.rets) a.rets[chunk] = (int64_t)(total_bytes / ESZ);   // element units, floor (:554)
    }
    return total_bytes;
}

template <int W, bool FIRE, bool EXACT, bool SPLIT = false, int DPT = 64, bool CM = false>
__global__ void __launch_bounds__(kThreads) encode_wide_kernel(EncodeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int DP = SPLIT ? 32 : DPT;
    constexpr uint32_t LOG2DP = DP == 4 ? 2 : DP == 8 ? 3 : DP == 16 ? 4 : DP == 32 ? 5 : 6;
    const uint32_t wg = workgroup_number(a.dn);
    const uint32_t size = encode_wide_body<W, FIRE, EXACT, SPLIT, DPT, CM>(a, wg);
    // the container, built before the workgroup leaves (compact_tail.h); without it the caller compacts the slots
    if (a.dn.dense) dense_tail(a.dn, wg, a.nchunks, LOG2DP, size, a.slots, a.slot_stride, smem, a.lds_group_stride);
}

}  // namespace sprintz

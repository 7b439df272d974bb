// decode_uni.h -- batched decoder for the LOW-DIM layout (sprintz_delta_lowdim.cpp:398-794 /
// sprintz_xff_lowdim.cpp:414-1119) with ND = 1 .. 4 columns (8 bits) / 1 or 2 (16 bits): the
// univariate streams of the paper's UCR archive and their few-column neighbours.  One lane per
// chunk (it carries all ND columns' predictor state).
//
// The generic kernel already maps a chunk of one column to one lane, but it walks memory a field
// at a time: with 64 K lanes each trailing its own stream and its own output, every dword read
// and every 8-byte write is a cache line of its own (0.1 - 0.17 TB/s).  Here, as in huf.hip's
// decoder:
//   * the stream waits in a per-lane LDS ring ring[dword][lane] (conflict-free for per-lane
//     cursors), refilled by 64-byte bursts on a fixed cadence with unconditionally issued loads;
//   * all lanes produce exactly one 8-sample block per step, so 64 bytes of output collect in
//     registers at compile-time positions, the four lanes of a quad transpose their 16-byte
//     pieces with DPP and every store writes one chunk's 64 contiguous bytes as one request.
#pragma once

#include "decode_kernel.h"
#include "decode_fast.h"

namespace sprintz {

#ifndef SPRINTZ_UNI_WINDOW_FAST
#define SPRINTZ_UNI_WINDOW_FAST 1
#endif

// Geometry shared by the kernel and its launcher.  A lane refills its ring once every R blocks (a 64-byte piece per
// refill, so R * STEPMAX <= 64); the ring holds RP pieces so that two refill periods fit behind the cursor's piece.
// 8-bit univariate streams (BASELINE config 1) take R = 4: one round of loads per 32 samples instead of per 8.
constexpr int decode_uni_stepmax(int W, int ND) { return (2 * ND * (W == 8 ? 3 : 4) + 7) / 8 + 2 + ND * W; }
// A piece the cursor makes room for is noticed at the next refill (<= R - 1 blocks later), requested there and parked one
// refill on: it is resident 2 R - 1 blocks after the cursor entered the piece before it, when the cursor is at most
// (2 R - 1) STEPMAX bytes in and the next block may look STEPMAX + 12 bytes ahead (a 64-bit payload window = three ring
// dwords).  Two pieces do while 2 R STEPMAX + 12 <= 64: R = 2 for 8-bit univariate streams (STEPMAX 11) -- the refill
// code runs in nearly every step it is placed in (one of 64 lanes always has room), so half as often is ~10 VALU and 8
// LDS writes a block less.
constexpr int decode_uni_refill_every(int W, int ND) { return (W == 8 && ND == 1) ? 2 : 1; }
constexpr int decode_uni_ring_pieces(int W, int ND)
{
    const int r = decode_uni_refill_every(W, ND), sm = decode_uni_stepmax(W, ND), hb = (2 * ND * (W == 8 ? 3 : 4) + 7) / 8;
    if (r == 1) return (2 * sm + hb + 6 <= 64) ? 2 : 4;
    return (2 * r * sm + 12 <= 64) ? 2 : 4;
}
constexpr int decode_uni_threads(int W, int ND) { return (decode_uni_ring_pieces(W, ND) == 4 && decode_uni_refill_every(W, ND) > 1) ? 128 : 256; }

// Q: query-on-compressed (decode_kernel.h): per-column max and sum ride along; reduce-only never stores samples
template <int W, bool FIRE, int ND = 1, int Q = 0>
__global__ void __launch_bounds__(decode_uni_threads(W, ND)) decode_uni_kernel(DecodeArgs a)
{
    constexpr int TPB = decode_uni_threads(W, ND);
    constexpr int RFE = decode_uni_refill_every(W, ND);
    constexpr int HB = Elem<W>::HB;
    constexpr int ESZ = W / 8;
    constexpr uint32_t MASK = Elem<W>::MASK;
    constexpr int BD = 2 * ESZ * ND;                       // dwords per block
    constexpr int WINS = (64 % (4 * BD) == 0) ? 1 : 3;     // 64-byte windows that hold a whole number of blocks (3 columns: 24-byte blocks)
    constexpr int BW = 16 * WINS / BD;                     // blocks per such span: 8 .. 2
    constexpr int HBYTES = (2 * ND * HB + 7) / 8;          // group header: 2 slots x ND fields of HB bits
    constexpr uint32_t STEPMAX = HBYTES + 2 + ND * W;      // most bytes one step takes: header + run length / payload
    // ring pieces of 64 bytes: a piece requested in one step is usable in the next, and a step that starts
    // up to STEPMAX - 1 bytes into a piece must not need the piece after the next resident one
    constexpr int RP = decode_uni_ring_pieces(W, ND);      // pieces in the ring (128 or 256 bytes per lane)
    static_assert(STEPMAX == (uint32_t)decode_uni_stepmax(W, ND) && RFE * STEPMAX <= 64, "refill period");
    constexpr uint32_t RDW = RP * 16;                      // ring dwords
    static_assert(BW * BD == 16 * WINS && BW >= 1, "whole blocks per span of windows");
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    typedef v4 __attribute__((aligned(1), may_alias)) v4a1;
    typedef uint32_t v2 __attribute__((ext_vector_type(2)));
    typedef v2 __attribute__((aligned(1), may_alias)) v2a1;

    __shared__ uint32_t ring[RDW * TPB];                   // 128 / 256 bytes per lane
    const int t = threadIdx.x;
    const uint64_t chunk = (uint64_t)blockIdx.x * TPB + t;
    const bool exists = chunk < a.nchunks;
    uint32_t* const my = ring + t;

    const uint64_t total = a.offsets[a.nchunks];
    const uint64_t off = exists ? a.offsets[chunk] : 0;
    const uint64_t stream_len = exists ? a.offsets[chunk + 1] - off : 0;
    // All stream loads go through ONE wave-uniform buffer descriptor from the 64-byte line of this wave's first stream to
    // the end of the container (decode_fast.h does the same): a piece is one 32-bit offset + immediates instead of four
    // clamped 64-bit addresses (~27 VALU a refill, and the refill code runs in nearly every step: some lane of the 64
    // always needs one), and a load that runs past the container returns 0 instead of faulting.
    const uint64_t chunk_w0 = (uint64_t)blockIdx.x * TPB + (uint64_t)(t & ~63);
    const uint64_t wave_base = wave_uniform64(a.offsets[chunk_w0 < a.nchunks ? chunk_w0 : 0] & ~(uint64_t)63);
    const uint64_t wave_span = ((total - wave_base) + 15) & ~(uint64_t)15;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.comp + wave_base), 0, (uint32_t)(wave_span < 0xffffffffull ? wave_span : 0xffffffffull), 0x00020000);
    const uint32_t rel0 = (uint32_t)((off & ~(uint64_t)63) - wave_base);      // this lane's piece 0 (meaningless for a lane without a chunk: it never asks)
    struct Piece { v4 v[4]; };
    auto load_piece = [&](uint32_t k, bool wanted) -> Piece {
        Piece pc;
        const uint32_t vo = wanted ? rel0 + (k << 6) : 0u;
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const auto tt = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo + 16u * m, 0, 0);
            pc.v[m] = v4{tt[0], tt[1], tt[2], tt[3]};
        }
        return pc;
    };
    auto park = [&](uint32_t k, const Piece& pc) {
        uint32_t* q = my + (k & (uint32_t)(RP - 1)) * (16u * TPB);
#pragma unroll
        for (int m = 0; m < 4; m++) {
            q[(4 * m + 0) * TPB] = pc.v[m].x;
            q[(4 * m + 1) * TPB] = pc.v[m].y;
            q[(4 * m + 2) * TPB] = pc.v[m].z;
            q[(4 * m + 3) * TPB] = pc.v[m].w;
        }
    };
#pragma unroll
    for (int k = 0; k < RP; k++) park((uint32_t)k, load_piece((uint32_t)k, exists));
    uint32_t fpiece = RP;
    Piece pend = {};
    bool have_pend = false;
    uint32_t c = (uint32_t)(off & 63);                     // byte cursor, from the start of piece0
    const uint32_t c_begin = c;
    // A step takes at most STEPMAX bytes and a piece is 64: when the cursor leaves a slot its
    // successors are full, and the refill requested in the next step is usable in the one after.
    auto refill = [&]() {
        if (have_pend) {
            park(fpiece, pend);
            fpiece++;
        }
        have_pend = exists && fpiece - (c >> 6) < (uint32_t)RP;
        if (have_pend) pend = load_piece(fpiece, true);    // (issued unconditionally -- an idle lane re-reading one hot line -- this cost four
                                                           //  wave-wide loads per block where one block in ~19 needs them: 0.426 -> see DESIGN.md)
    };
    auto rd_dw = [&](uint32_t dw) -> uint32_t { return my[(dw & (RDW - 1u)) * TPB]; };
    auto rd8 = [&](uint32_t at) -> uint32_t { return (rd_dw(at >> 2) >> ((at & 3u) * 8u)) & 0xffu; };
    auto rd32 = [&](uint32_t at) -> uint32_t { return __builtin_amdgcn_alignbyte(rd_dw((at >> 2) + 1u), rd_dw(at >> 2), at); };

    // ---- 8-byte stream header (format.h:48-62)
    uint32_t groups_left = 0, remaining = 0;
    bool corrupt = false, alive = exists;
    if (exists) {
        const uint32_t w0 = rd32(c), w1 = rd32(c + 4);
        groups_left = w0;
        remaining = w1 & 0xffffu;
        c += 8;
        corrupt = (w1 >> 16) != (uint32_t)ND || groups_left > a.chunk_len / (8u * ND) + 2u || stream_len < 8;
        if (corrupt) { groups_left = 0; alive = false; }
    }

    uint32_t pv[ND];
    int pd[ND], ctr[ND];
    uint32_t nb0[ND], nb1[ND];
#pragma unroll
    for (int k = 0; k < ND; k++) { pv[k] = 0; pd[k] = 0; ctr[k] = 0; nb0[k] = 0; nb1[k] = 0; }
    uint32_t run_left = 0, out_elems = 0;
    uint32_t qmax[ND];
    uint64_t qsum[ND];
#pragma unroll
    for (int k = 0; k < ND; k++) { qmax[k] = 0; qsum[k] = 0; }
    int slot = 2;
    uint8_t* const obase = (uint8_t*)a.out + chunk * (uint64_t)a.chunk_len * ESZ;
    const bool odd1 = (t & 1) != 0, odd2 = (t & 2) != 0;
    const uint32_t part = (uint32_t)t & 3u;

    for (;;) {
        uint32_t win[BW][BD];                              // the span of 64-byte windows: block b at compile-time position b
        uint32_t valid = 0;                                // bit b: block b of the window was produced
        const uint32_t win_elems = out_elems;              // output position of the window's first block
        bool wave_done = false;
        // Round 3, univariate streams: the WINDOW-level fast path.  When every lane of the wave is alive, outside a run, at the same
        // slot, and has room for a whole window (eight blocks of output, eight steps of stream, four group headers), none of that
        // needs testing block by block: a block is then "read the header byte" (slot 0 only), "is the width zero anywhere?" (one
        // ballot: a run or padding slot sends this block, and the rest of the window, down the general path) and four additions.
        // Chunks start in phase and stay in phase until one of them meets a run, so this is the normal case.
        bool window_fast = false;
        int sslot = 0;
        if constexpr (ND == 1 && SPRINTZ_UNI_WINDOW_FAST) {
            sslot = __builtin_amdgcn_readfirstlane(slot);
            const bool ok = alive && run_left == 0 && (slot == 1 || slot == 2) && slot == sslot && groups_left >= (uint32_t)(BW / 2 + 1) &&
                            out_elems + 8u * BW <= a.chunk_len && (uint64_t)(c - c_begin) + (uint64_t)BW * STEPMAX <= stream_len + 2;
            window_fast = __ballot(!ok) == 0 && __ballot(true) == ~0ull;
        }
#pragma unroll
        for (int b = 0; b < BW; b++) {
#pragma unroll
            for (int d = 0; d < BD; d++) win[b][d] = 0;
            if (b % RFE == 0) refill();
            // ---- this lane's next block: inside a run, or the next slot of the stream
            bool have = false;
            uint32_t nb[ND], cfield = 0, nbsum = 0;
#pragma unroll
            for (int k = 0; k < ND; k++) nb[k] = 0;
            // ---- univariate streams: the common step (inside a run, or a payload block, with or without a new group
            // header in front of it) is taken by the whole wave at once with selects instead of divergent branches --
            // the general state machine below costs ~130 scalar instructions of exec-mask bookkeeping per block.
            // It still takes every step in which some lane meets a run length, a padding slot or the stream's end.
            bool fast_step = false;
            if constexpr (ND == 1 && SPRINTZ_UNI_WINDOW_FAST) {
                if (window_fast) {
                    if (sslot == 2) {                                           // a group header, then its slot 0
                        const uint32_t h = rd8(c);
                        const uint32_t f0 = h & ((1u << HB) - 1u), f1 = (h >> HB) & ((1u << HB) - 1u);
                        const uint32_t n0 = f0 == (uint32_t)(W - 1) ? (uint32_t)W : f0, n1 = f1 == (uint32_t)(W - 1) ? (uint32_t)W : f1;
                        if (__ballot(n0 == 0u) != 0) {
                            window_fast = false;                                // somebody's slot 0 is a run or padding: nothing taken yet
                        } else {
                            groups_left--;
                            nb0[0] = n0; nb1[0] = n1;
                            nb[0] = n0; nbsum = n0;
                            cfield = c + (uint32_t)HBYTES;
                            c += (uint32_t)HBYTES + n0;
                            slot = 1; sslot = 1;
                            have = true; fast_step = true;
                        }
                    } else {                                                    // slot 1 of the group the lane is in
                        if (__ballot(nb1[0] == 0u) != 0) {
                            window_fast = false;
                        } else {
                            nb[0] = nb1[0]; nbsum = nb1[0];
                            cfield = c;
                            c += nb1[0];
                            slot = 2; sslot = 2;
                            have = true; fast_step = true;
                        }
                    }
                }
            }
            if constexpr (ND == 1) {
                if (!fast_step) {
                const bool in_run = alive && run_left > 0;
                const bool need_hdr = alive && !in_run && slot == 2;
                uint32_t hpeek = 0;
                if (__ballot(need_hdr) != 0) hpeek = rd8(c);
                const uint32_t f0 = hpeek & ((1u << HB) - 1u), f1 = (hpeek >> HB) & ((1u << HB) - 1u);
                const uint32_t n0 = need_hdr ? (f0 == (uint32_t)(W - 1) ? (uint32_t)W : f0) : nb0[0];
                const uint32_t n1 = need_hdr ? (f1 == (uint32_t)(W - 1) ? (uint32_t)W : f1) : nb1[0];
                const uint32_t s_eff = need_hdr ? 0u : (uint32_t)slot;
                const uint32_t nbn = s_eff ? n1 : n0;                       // width of the slot this step would take
                const bool simple = !alive || in_run || (s_eff < 2u && nbn != 0u && (!need_hdr || groups_left > 0u));
                if (__ballot(!simple) == 0) {
                    fast_step = true;
                    const bool take = alive && !in_run;
                    groups_left -= need_hdr ? 1u : 0u;
                    c += need_hdr ? (uint32_t)HBYTES : 0u;
                    nb0[0] = n0;
                    nb1[0] = n1;
                    run_left -= in_run ? 1u : 0u;
                    nb[0] = take ? nbn : 0u;
                    nbsum = nb[0];
                    cfield = c;
                    c += nb[0];
                    slot = take ? (int)s_eff + 1 : slot;
                    have = alive;
                    const bool over = alive && ((uint64_t)(c - c_begin) > stream_len + 2 || out_elems + 8 > a.chunk_len);
                    corrupt = corrupt || over;
                    alive = alive && !over;
                    have = have && !over;
                }
                }
            }
            if (alive && !fast_step) {
                if (run_left > 0) {
                    run_left--;
                    have = true;
                } else {
                    for (int tries = 0; tries < 4 && !have && alive; tries++) {   // <= 2 padding slots in a row in a valid stream
                        if (slot == 2) {
                            if (groups_left == 0) { alive = false; break; }
                            groups_left--;
                            const uint32_t h = rd32(c);                         // 2 x ND fields of HB bits (:713-735)
                            c += HBYTES;
#pragma unroll
                            for (int k = 0; k < ND; k++) {
                                const uint32_t f0 = (h >> (k * HB)) & ((1u << HB) - 1u), f1 = (h >> ((ND + k) * HB)) & ((1u << HB) - 1u);
                                nb0[k] = f0 == (uint32_t)(W - 1) ? (uint32_t)W : f0;
                                nb1[k] = f1 == (uint32_t)(W - 1) ? (uint32_t)W : f1;
                            }
                            slot = 0;
                        }
                        uint32_t sum = 0;
#pragma unroll
                        for (int k = 0; k < ND; k++) { nb[k] = slot ? nb1[k] : nb0[k]; sum += nb[k]; }
                        slot++;
                        if (sum == 0) {                                         // RUN slot: varint length in blocks
                            const uint32_t b0 = rd8(c);
                            uint32_t len = b0 & 0x7fu;
                            c += 1;
                            if (b0 & 0x80u) { len |= rd8(c) << 7; c += 1; }
                            if (len > 0) { run_left = len - 1; have = true; }
                        } else {
                            nbsum = sum;
                            cfield = c;
                            c += sum;                                           // per column 8 fields of nb bits = nb bytes
                            have = true;
                        }
                    }
                    if (!have && alive) { corrupt = true; alive = false; }
                }
                if ((uint64_t)(c - c_begin) > stream_len + 2 || (have && out_elems + 8 * ND > a.chunk_len)) {
                    corrupt = true; alive = false; have = false;
                }
            }
            if (__ballot(alive || have) == 0) { wave_done = true; }
            // ---- the block: zigzag^-1 + forecast recurrence down the column
            if (have) {
                uint32_t x[ND][8];
                uint32_t cf = cfield;
#pragma unroll
                for (int k = 0; k < ND; k++) {
                    const int coef = FIRE ? fire_coef<W, true>(ctr[k]) : 0;
                    const uint32_t nbk = nbsum ? nb[k] : 0u;                 // inside a run every field is empty
                    int grad = 0;
                    // W == 8: a column's 8 fields are nbk <= 8 contiguous bytes -- ONE 64-bit window per column (three aligned
                    // ring dwords, two v_alignbyte), its fields taken four at a time (4 nbk <= 32 bits: v_alignbit + v_bfe each)
                    uint32_t plo = 0, phi = 0;
                    uint32_t w1 = 0, wm = 0, s1 = 0, s2 = 0, s3 = 0;
                    if constexpr (W == 8) {
                        const uint32_t d0 = rd_dw(cf >> 2), d1 = rd_dw((cf >> 2) + 1u), d2 = rd_dw((cf >> 2) + 2u);
                        plo = __builtin_amdgcn_alignbyte(d1, d0, cf);
                        phi = __builtin_amdgcn_alignbyte(d2, d1, cf);
                        // zigzag^-1 fused with the field fetch, as in decode_fast.h: err = bfe_u(w, s + 1, nb - 1) ^ bfe_i(w, s, 1)
                        w1 = nbk != 0u ? 1u : 0u;                 // width of the sign field
                        wm = nbk - w1;                            // width of the magnitude field
                        s1 = nbk; s2 = 2u * nbk; s3 = 3u * nbk;   // four fields are 4 nbk <= 32 bits: all inside one dword
                    }
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        int err;
                        if constexpr (W == 8) {
                            if (i == 4) {                         // the upper four fields start at bit 4 nbk
                                const uint64_t both = (((uint64_t)phi << 32) | plo) >> ((4u * nbk) & 63u);
                                plo = (uint32_t)both;
                            }
                            const uint32_t sh = (i & 3) == 0 ? 0u : (i & 3) == 1 ? s1 : (i & 3) == 2 ? s2 : s3;
                            err = (int)(__builtin_amdgcn_ubfe(plo, sh + 1u, wm) ^ (uint32_t)__builtin_amdgcn_sbfe((int)plo, sh, w1));
                        } else {
                            uint32_t z = 0;
                            if (nbk != 0) {
                                const uint32_t bit = (cf & 3u) * 8u + (uint32_t)i * nbk;
                                const uint32_t dw = (cf >> 2) + (bit >> 5);
                                z = __builtin_amdgcn_ubfe(__builtin_amdgcn_alignbit(rd_dw(dw + 1u), rd_dw(dw), bit & 31u), 0, nbk);
                            }
                            err = unzigzag(z);
                        }
                        const int pred = FIRE ? fire_predict<W, true>(pd[k], coef) : 0;
                        const int delta = sext<W>(err + pred);
                        if (FIRE && (i & 1)) grad += sign_times(err, pd[k]);
                        if constexpr (W == 8 && ND == 1 && !FIRE && Q == 0) {
                            pv[k] += (uint32_t)err;               // only the low byte is ever used (v_perm packing below): no mask, no sign extension
                        } else {
                            pv[k] = (pv[k] + (uint32_t)delta) & MASK;
                        }
                        pd[k] = delta;
                        x[k][i] = pv[k];
                        if constexpr (Q != 0) { qmax[k] = pv[k] > qmax[k] ? pv[k] : qmax[k]; qsum[k] += pv[k]; }
                    }
                    if (FIRE && nbsum != 0) ctr[k] = wrap_counter<W>(ctr[k] + (sext<W>(grad) >> 2));   // counters only move on real blocks
                    cf += nbk;
                }
                // row-major block: element e = row * ND + column at byte e * ESZ
                if constexpr (W == 8 && ND == 1) {             // four low bytes -> one dword: three v_perm_b32, no masks
                    const uint32_t p01 = __builtin_amdgcn_perm(x[0][1], x[0][0], 0x0c0c0400u), p23 = __builtin_amdgcn_perm(x[0][3], x[0][2], 0x0c0c0400u);
                    const uint32_t p45 = __builtin_amdgcn_perm(x[0][5], x[0][4], 0x0c0c0400u), p67 = __builtin_amdgcn_perm(x[0][7], x[0][6], 0x0c0c0400u);
                    win[b][0] = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
                    win[b][1] = __builtin_amdgcn_perm(p67, p45, 0x05040100u);
                } else {
#pragma unroll
                for (int e = 0; e < 8 * ND; e++) win[b][(e * ESZ) / 4] |= x[e % ND][e / ND] << (((e * ESZ) % 4) * 8);
                }
                valid |= 1u << b;
                out_elems += 8 * ND;
            }
        }
        // ---- the window leaves.  Full windows: the quad transposes its 16-byte pieces (two DPP
        // butterfly stages: (member m, piece k) -> (lane k, slot m)) and stores one member's 64
        // bytes per instruction.  A partly filled window (the chunk's end) is stored block by block.
        const bool full = valid == (1u << BW) - 1u;
        if constexpr (Q != kQueryReduceOnly) {
#pragma unroll
        for (int wn = 0; wn < WINS; wn++) {
        uint32_t v[4][4];
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int d = 0; d < 4; d++) v[k][d] = win[(16 * wn + 4 * k + d) / BD][(16 * wn + 4 * k + d) % BD];
#pragma unroll
        for (int k = 0; k < 4; k += 2)
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const uint32_t send = odd1 ? v[k][d] : v[k + 1][d];
                const uint32_t recv = (uint32_t)__builtin_amdgcn_mov_dpp((int)send, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
                if (odd1) v[k][d] = recv; else v[k + 1][d] = recv;
            }
#pragma unroll
        for (int k = 0; k < 2; k++)
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const uint32_t send = odd2 ? v[k][d] : v[k + 2][d];
                const uint32_t recv = (uint32_t)__builtin_amdgcn_mov_dpp((int)send, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
                if (odd2) v[k][d] = recv; else v[k + 2][d] = recv;
            }
        const uint64_t mine = full ? (uint64_t)(uintptr_t)(obase + (uint64_t)win_elems * ESZ + 64u * wn) : 0ull;
        const int mlo = (int)(uint32_t)mine, mhi = (int)(uint32_t)(mine >> 32);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t dlo, dhi;
            if (q == 0) { dlo = (uint32_t)__builtin_amdgcn_mov_dpp(mlo, 0x00, 0xf, 0xf, true); dhi = (uint32_t)__builtin_amdgcn_mov_dpp(mhi, 0x00, 0xf, 0xf, true); }
            else if (q == 1) { dlo = (uint32_t)__builtin_amdgcn_mov_dpp(mlo, 0x55, 0xf, 0xf, true); dhi = (uint32_t)__builtin_amdgcn_mov_dpp(mhi, 0x55, 0xf, 0xf, true); }
            else if (q == 2) { dlo = (uint32_t)__builtin_amdgcn_mov_dpp(mlo, 0xAA, 0xf, 0xf, true); dhi = (uint32_t)__builtin_amdgcn_mov_dpp(mhi, 0xAA, 0xf, 0xf, true); }
            else { dlo = (uint32_t)__builtin_amdgcn_mov_dpp(mlo, 0xFF, 0xf, 0xf, true); dhi = (uint32_t)__builtin_amdgcn_mov_dpp(mhi, 0xFF, 0xf, 0xf, true); }
            const uint64_t dst = ((uint64_t)dhi << 32) | dlo;
#ifdef SPRINTZ_ABL_UNI_NO_STORE
            if (dst == 1) {
#else
            if (dst) {
#endif
                v4 piece = {v[q][0], v[q][1], v[q][2], v[q][3]};
                __builtin_nontemporal_store(piece, (v4a1*)(uintptr_t)(dst + 16u * part));   // written once, never re-read here: 0.402 -> 0.336 ms on config 1
            }
        }
        }
        if (!full && valid) {
            uint8_t* d = obase + (uint64_t)win_elems * ESZ;
#pragma unroll
            for (int b = 0; b < BW; b++) {
                if ((valid >> b) & 1u) {                   // blocks are produced in order: the valid ones are a prefix
                    if constexpr (BD == 2) { v2 p = {win[b][0], win[b][1]}; *(v2a1*)(d + 8 * b) = p; }
                    else if constexpr (BD % 4 != 0) {
#pragma unroll
                        for (int j = 0; j < BD; j += 2) { v2 p = {win[b][j], win[b][j + 1]}; *(v2a1*)(d + 4 * BD * b + 4 * j) = p; }
                    } else {
#pragma unroll
                        for (int j = 0; j < BD; j += 4) { v4 p = {win[b][j], win[b][j + 1], win[b][j + 2], win[b][j + 3]}; *(v4a1*)(d + 4 * BD * b + 4 * j) = p; }
                    }
                }
            }
        }
        }   // Q != kQueryReduceOnly
        if (wave_done) break;
    }

    // ---- verbatim tail (:1171), straight from HBM
    if (exists) {
        if (!corrupt && out_elems + remaining > a.chunk_len) corrupt = true;
        if (!corrupt && (uint64_t)(c - c_begin) + (uint64_t)remaining * ESZ > stream_len + 2) corrupt = true;
        if (!corrupt) {
            const uint8_t* src = a.comp + off + (c - c_begin);
            if constexpr (Q != kQueryReduceOnly) {
                uint8_t* d = obase + (uint64_t)out_elems * ESZ;
                for (uint32_t j = 0; j < remaining * ESZ; j++) d[j] = src[j];
            }
            if constexpr (Q != 0) {                      // the verbatim tail continues the row-major order: element e is column e % ND
                for (uint32_t e = 0; e < remaining; e++) {
                    const uint32_t x = ESZ == 1 ? (uint32_t)src[e] : ((uint32_t)src[2 * e] | ((uint32_t)src[2 * e + 1] << 8));
#pragma unroll
                    for (int k = 0; k < ND; k++)
                        if (e % ND == (uint32_t)k) { qmax[k] = x > qmax[k] ? x : qmax[k]; qsum[k] += x; }
                }
#pragma unroll
                for (int k = 0; k < ND; k++)
                    if (a.qres) a.qres[chunk * (uint64_t)ND + (uint64_t)k] = a.qop == 1 ? (uint64_t)qmax[k] : qsum[k];
            }
        }
        if (a.rets) a.rets[chunk] = corrupt ? kErrCorrupt : (int64_t)out_elems + remaining;
    }
}

}  // namespace sprintz

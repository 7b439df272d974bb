// encode_uni.h -- batched encoder for the LOW-DIM layout (sprintz_delta_lowdim.cpp:39-384 /
// sprintz_xff_lowdim.cpp:44-400) with ND = 1 .. 4 columns (8 bits) / 1 or 2 (16 bits).  One lane
// per chunk; same stream bytes as encode_kernel.h, which it follows step for step (the RLE state
// machine of SURVEY.md A.5 included).
//
// Why a kernel of its own: see decode_uni.h -- a lane per chunk walking memory 8 bytes at a time
// makes every access a cache line of its own.  Here every lane consumes exactly one 8-sample
// block per step, so
//   * INPUT arrives 64 bytes per lane at a time: the four lanes of a quad load one member's 64
//     bytes as ONE request (16 bytes each), one window ahead, and a DPP 4 x 4 transpose hands
//     every lane its own window in registers, at compile-time positions;
//   * OUTPUT (variable length) is OR-ed into a 128-byte per-lane LDS ring out[dword][lane] -- the
//     group header stays patchable until its second slot is known -- and leaves in 64-byte units,
//     four 16-byte stores back to back.
#pragma once

#include "encode_kernel.h"
#include "decode_fast.h"

namespace sprintz {

template <int W, bool FIRE, int ND = 1>
__global__ void __launch_bounds__(256) encode_uni_kernel(EncodeArgs a)
{
    constexpr int HB = Elem<W>::HB;
    constexpr int ESZ = W / 8;
    constexpr int BB = 8 * ESZ * ND;                       // bytes per block
    constexpr int WINS = (64 % BB == 0) ? 1 : 3;           // 64-byte input windows that hold a whole number of blocks (3 columns: 24-byte blocks)
    constexpr int BW = 64 * WINS / BB;                     // blocks per such span: 8 .. 2
    constexpr int HBYTES = (2 * ND * HB + 7) / 8;          // group header: 2 slots x ND fields of HB bits
    constexpr uint32_t GROUPMAX = HBYTES + 2 * ND * W + 4; // most bytes between two flushes: header, two blocks, close-out
    constexpr uint32_t RDW = (63 + GROUPMAX <= 128) ? 32 : 64;   // ring dwords per lane (128 or 256 bytes)
    constexpr uint32_t RM = RDW - 1;
    static_assert(BW >= 1 && BW * BB == 64 * WINS, "whole blocks per span of windows");
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    typedef v4 __attribute__((aligned(1), may_alias)) v4a1;

    __shared__ uint32_t oring[RDW * 256];                  // 128 / 256 bytes of output per lane
    const int t = threadIdx.x;
    const uint64_t chunk = (uint64_t)blockIdx.x * 256 + t;
    const bool exists = chunk < a.nchunks;
    uint32_t* const my = oring + t;
#pragma unroll
    for (int d = 0; d < (int)RDW; d++) my[d * 256] = 0;

    const uint64_t first = chunk * (uint64_t)a.chunk_len;
    const uint32_t n = exists ? (uint32_t)((a.total_len - first < a.chunk_len) ? (a.total_len - first) : a.chunk_len) : 0u;
    const uint8_t* const sc = (const uint8_t*)a.src + first * ESZ;
    uint8_t* const gdst = a.slots + chunk * a.slot_stride;
    const uint32_t nbytes = n * ESZ;

    // ---- output ring: stream bytes [flushed, flushed + 4 * RDW), flushed a multiple of 64
    uint32_t wpos = a.write_size ? 8u : 0u;
    uint32_t flushed = 0;
    auto or_bits = [&](uint32_t bp, uint32_t v, uint32_t nb) {           // low nb (<= 16) bits of v at stream bit bp
        const uint32_t sh = bp & 31u;
        uint32_t* q = my + (((bp >> 5) & RM) << 8);
        q[0] |= v << sh;
        if (sh + nb > 32u) my[(((bp >> 5) + 1u) & RM) << 8] |= v >> (32u - sh);
    };
    auto put_byte = [&](uint32_t pos, uint32_t v) { my[((pos >> 2) & RM) << 8] |= v << ((pos & 3u) * 8u); };
    auto flush_to = [&](uint32_t upto) {                                 // whole 64-byte units below upto -> HBM, re-zeroed
        while (flushed < upto) {
            const uint32_t d0 = (flushed >> 2) & RM;                     // a multiple of 16
            v4 p[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                p[q] = v4{my[(d0 + 4 * q) << 8], my[(d0 + 4 * q + 1) << 8], my[(d0 + 4 * q + 2) << 8], my[(d0 + 4 * q + 3) << 8]};
            }
#pragma unroll
            for (int d = 0; d < 16; d++) my[(d0 + d) << 8] = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) *(v4a1*)(gdst + flushed + 16 * q) = p[q];
            flushed += 64;
        }
    };
    auto put_run = [&](uint32_t run) {                                   // sprintz_xff_lowdim.cpp:246-253
        put_byte(wpos, (run & 0x7fu) | (run > 0x7fu ? 0x80u : 0u));
        if (run > 0x7fu) put_byte(wpos + 1, run >> 7);
        wpos += run > 0x7fu ? 2u : 1u;
    };

    constexpr uint32_t BLK = 8 * ND;                                     // elements per block
    const int64_t limit = (int64_t)n - 2 * (int64_t)BLK;                 // last_full_group_start
    int64_t pos_in = 0;
    uint32_t ngroups = 0, run = 0, hdr_pos = 0;
    int slot = 0;
    uint32_t pv[ND];
    int pd[ND], ctr[ND];
#pragma unroll
    for (int k = 0; k < ND; k++) { pv[k] = 0; pd[k] = 0; ctr[k] = 0; }
    uint32_t hdr = 0;                                                    // this group's header (<= 24 bits), written when it closes
    auto put_hdr = [&]() {
#pragma unroll
        for (int k = 0; k < HBYTES; k++) put_byte(hdr_pos + k, (hdr >> (8 * k)) & 0xffu);
    };
    auto start_group = [&]() {
        if (ngroups) put_hdr();
        hdr = 0;
        ngroups++;
        flush_to(wpos & ~63u);
        hdr_pos = wpos;
        wpos += HBYTES;                                                  // 2 x ND fields of HB bits
        slot = 0;
    };
    bool active = exists && n >= 128u;
    if (active) start_group();

    // ---- input windows: a quad loads one member's 64 bytes per instruction, one window ahead
    const bool odd1 = (t & 1) != 0, odd2 = (t & 2) != 0;
    const uint32_t part = (uint32_t)t & 3u;
    const uint64_t sa = (uint64_t)(uintptr_t)sc;
    const int salo = (int)(uint32_t)sa, sahi = (int)(uint32_t)(sa >> 32);
    auto bcast = [&](int x, int q) -> uint32_t {
        if (q == 0) return (uint32_t)__builtin_amdgcn_mov_dpp(x, 0x00, 0xf, 0xf, true);
        if (q == 1) return (uint32_t)__builtin_amdgcn_mov_dpp(x, 0x55, 0xf, 0xf, true);
        if (q == 2) return (uint32_t)__builtin_amdgcn_mov_dpp(x, 0xAA, 0xf, 0xf, true);
        return (uint32_t)__builtin_amdgcn_mov_dpp(x, 0xFF, 0xf, 0xf, true);
    };
#ifndef SPRINTZ_ENCUNI_PAIR
#define SPRINTZ_ENCUNI_PAIR 1
#endif
    // SPRINTZ_ENCUNI_PAIR: the two 64-byte windows of a 128-byte line are requested TOGETHER (two windows ahead / one ahead) instead of one per
    // window: requested a window apart, the line had left the L2 again before its second half was asked for (FETCH_SIZE x 2 = 923 MB for
    // 537 MB of samples on BASELINE config 1)
    uint32_t nxtA[WINS][4][4], nxtB[WINS][4][4];
    auto load_window = [&](uint32_t wi, uint32_t (&nxt)[WINS][4][4]) {   // nxt[.][q] = part `part` of member q's windows WINS*wi ..
#pragma unroll
        for (int wn = 0; wn < WINS; wn++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint64_t base = ((uint64_t)bcast(sahi, q) << 32) | bcast(salo, q);
                const uint32_t nb_q = bcast((int)nbytes, q);
                const uint32_t o = (wi * WINS + wn) * 64u + part * 16u;
                v4 x = {0, 0, 0, 0};
                if (o < nb_q) x = *(const v4a1*)(uintptr_t)(base + o);   // may run <= 15 bytes past the chunk (READ_SLACK)
                nxt[wn][q][0] = x.x; nxt[wn][q][1] = x.y; nxt[wn][q][2] = x.z; nxt[wn][q][3] = x.w;
            }
    };
    // (member m, piece k) -> (lane k, slot m): two DPP butterfly stages, then this lane owns its window
    auto take_window = [&](const uint32_t (&nxt)[WINS][4][4], uint32_t (&vv)[WINS][4][4]) {
#pragma unroll
        for (int wn = 0; wn < WINS; wn++) {
        uint32_t (&v)[4][4] = vv[wn];
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int d = 0; d < 4; d++) v[q][d] = nxt[wn][q][d];
#pragma unroll
        for (int k = 0; k < 4; k += 2)
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const uint32_t send = odd1 ? v[k][d] : v[k + 1][d];
                const uint32_t recv = (uint32_t)__builtin_amdgcn_mov_dpp((int)send, 0xB1, 0xf, 0xf, true);
                if (odd1) v[k][d] = recv; else v[k + 1][d] = recv;
            }
#pragma unroll
        for (int k = 0; k < 2; k++)
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const uint32_t send = odd2 ? v[k][d] : v[k + 2][d];
                const uint32_t recv = (uint32_t)__builtin_amdgcn_mov_dpp((int)send, 0x4E, 0xf, 0xf, true);
                if (odd2) v[k][d] = recv; else v[k + 2][d] = recv;
            }
        }
    };
    auto code_window = [&](const uint32_t (&vv)[WINS][4][4]) {
#pragma unroll
        for (int b = 0; b < BW; b++) {
            if (!active) continue;
            // ---- the block at pos_in (== (wi * BW + b) * 8 * ND): forecast + zigzag + OR-mask (sprintz_xff_lowdim.cpp:160-215)
            uint32_t z[ND][8], nb[ND], total = 0;
#pragma unroll
            for (int k = 0; k < ND; k++) {
                uint32_t mask = 0;
                const int coef = FIRE ? fire_coef<W, true>(ctr[k]) : 0;
                int grad = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int byte = (b * 8 * ND + i * ND + k) * ESZ;        // row-major element (row i, column k) inside the span
                    const uint32_t x = (vv[byte / 64][(byte % 64) / 16][(byte % 16) / 4] >> (8 * (byte % 4))) & Elem<W>::MASK;
                    const int delta = sext<W>((int)(x - pv[k]));
                    const int pred = FIRE ? fire_predict<W, true>(pd[k], coef) : 0;
                    const int err = sext<W>(delta - pred);
                    const uint32_t zz = zigzag<W>(err);
                    if (FIRE && (i & 1)) grad += sign_times(err, pd[k]);
                    mask |= zz;
                    z[k][i] = zz;
                    pv[k] = x;
                    pd[k] = delta;
                }
                if (FIRE) ctr[k] = wrap_counter<W>(ctr[k] + (sext<W>(grad) >> 2));
                nb[k] = nbits_of<W, true>(mask);
                total += nb[k];
            }

            // ---- RLE state machine ("<" tail test: sprintz_delta_lowdim.cpp:190, sprintz_xff_lowdim.cpp:234)
            for (;;) {
                if (total == 0 && run < 0x7fffu) {
                    run++;
                    pos_in += BLK;
                    if (pos_in < limit) break;                           // analyse the next block
                    slot++;                                              // not enough input left: close the run
                    put_run(run);
                    wpos += (uint32_t)(2 - slot);                        // empty slots are one 0x00 byte each (ring is zero)
                    run = 0;
                    active = false;
                    break;
                }
                if (run > 0) {                                           // a run just ended
                    slot++;
                    put_run(run);
                    run = 0;
                    if (slot == 2) start_group();
                    continue;                                            // re-evaluate this block
                }
#pragma unroll
                for (int k = 0; k < ND; k++) {
                    const uint32_t nbk = nb[k];
                    const uint32_t f = nbk == (uint32_t)W ? (uint32_t)(W - 1) : nbk;
                    hdr |= f << ((uint32_t)(slot * ND + k) * HB);
                    if (nbk == 0) continue;
                    // 8 fields of nbk bits = nbk bytes, assembled in registers, OR-ed in dword by dword
                    uint64_t lo = 0, hi = 0;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        lo |= (uint64_t)z[k][i] << (i * nbk);
                        hi |= (uint64_t)z[k][4 + i] << (i * nbk);
                    }
                    const uint32_t half = 4u * nbk;                      // 4 .. 64 bits
                    uint32_t acc[5];
                    if constexpr (W == 8) {                              // half <= 32
                        lo |= hi << half;
                        acc[0] = (uint32_t)lo; acc[1] = (uint32_t)(lo >> 32); acc[2] = acc[3] = acc[4] = 0;
                    } else {
                        const uint64_t top = half == 64u ? hi : (hi >> (64u - half));
                        if (half != 64u) lo |= hi << half;
                        acc[0] = (uint32_t)lo; acc[1] = (uint32_t)(lo >> 32); acc[2] = (uint32_t)top; acc[3] = (uint32_t)(top >> 32); acc[4] = 0;
                    }
                    const uint32_t sh = (wpos & 3u) * 8u, d0 = wpos >> 2, span = (wpos & 3u) + nbk;   // bytes from dword d0 on
                    constexpr int NDW = W == 8 ? 3 : 5;
                    uint32_t prev = 0;
#pragma unroll
                    for (int q = 0; q < NDW; q++) {
                        const uint32_t cur = q < (W == 8 ? 2 : 4) ? acc[q] : 0u;
                        const uint32_t dw = (uint32_t)(((((uint64_t)cur << 32) | prev) << sh) >> 32);
                        if ((uint32_t)(4 * q) < span) my[((d0 + q) & RM) << 8] |= dw;
                        prev = cur;
                    }
                    wpos += nbk;
                }
                pos_in += BLK;
                slot++;
                if (slot == 2) {
                    if (pos_in <= limit) start_group();
                    else active = false;
                }
                break;
            }
        }
    };
    load_window(0, nxtA);
    // (three columns: a span is three windows and the second buffer costs 48 registers -- one wave a SIMD, 0.30 -> 0.50 ms: not there)
    if constexpr (SPRINTZ_ENCUNI_PAIR && WINS == 1) {
        load_window(1, nxtB);
        for (uint32_t wi = 0;; wi += 2) {
            if (__ballot(active) == 0) break;
            uint32_t vv[WINS][4][4];
            take_window(nxtA, vv);
            code_window(vv);
            take_window(nxtB, vv);
            load_window(wi + 2, nxtA);                                   // the next line's two halves, back to back
            load_window(wi + 3, nxtB);
            code_window(vv);
        }
    } else {
        for (uint32_t wi = 0;; wi++) {
            if (__ballot(active) == 0) break;
            uint32_t vv[WINS][4][4];
            take_window(nxtA, vv);
            load_window(wi + 1, nxtA);
            code_window(vv);
        }
    }

    if (!exists) return;
    if (ngroups) put_hdr();
    // ---- verbatim tail (sprintz_xff_lowdim.cpp:398) through the ring, 16 source bytes at a time
    const uint32_t remaining = (uint32_t)((int64_t)n - pos_in);
    {
        const uint8_t* const tp = sc + (size_t)pos_in * ESZ;
        const uint32_t left = remaining * ESZ;
        for (uint32_t o = 0; o < left; o += 16) {
            flush_to(wpos & ~63u);                                       // >= 64 bytes of room
            const v4 x = *(const v4a1*)(tp + o);                         // starts inside the chunk (READ_SLACK covers the end)
            const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int r = (int)(left - o) - 4 * q;                   // tail bytes from this dword on
                if (r > 0) or_bits((wpos + 4u * q) * 8u, r >= 4 ? xs[q] : xs[q] & ((1u << (8 * r)) - 1u), 32u);
            }
            wpos += left - o < 16u ? left - o : 16u;
        }
    }
    flush_to(wpos & ~63u);
    for (uint32_t p = flushed; p < wpos; p += 16) {                      // the last units, rounded up to 16 bytes (slot slack)
        const uint32_t d0 = (p >> 2) & RM;
        *(v4a1*)(gdst + p) = v4{my[d0 << 8], my[(d0 + 1) << 8], my[(d0 + 2) << 8], my[(d0 + 3) << 8]};
    }
    if (a.write_size) {                                                  // format.h:36-45
        ((uint32_t*)gdst)[0] = ngroups;
        ((uint32_t*)gdst)[1] = (remaining & 0xffffu) | ((uint32_t)ND << 16);
    }
    a.sizes[chunk] = wpos;
    if (a.rets) a.rets[chunk] = (int64_t)(wpos / ESZ);
}

}  // namespace sprintz

// encode_lat.h -- the LATENCY encoder: one WORKGROUP (256 lanes) per chunk, the counterpart of decode_lat.h for batches too
// small to fill the chip (a single drop-in call above all: encode_fast.h's lane group takes 90 us for one 10 KB chunk).  Same
// stream bytes, sizes and return values as encode_fast.h / encode_kernel.h (sprintz_xff_rle.cpp:61-555,
// sprintz_delta_rle.cpp:55-404; the low-dim layouts as template parameter LOW), ndims <= 64, chunks of at most 16 KB.
//
// What is serial in the encoder is little:
//   (1) the forecast's coefficient: coef[b+1] depends on the gradient of block b, i.e. on the signs of block b's errors in
//       its four odd rows (:240-241, :273-275) -- a chain of ~20 instructions a block when four lanes share a column;
//   (2) the RLE / group state machine over the blocks' "all zero?" bits and sizes (:350-456) -- ~15 scalar
//       instructions a block, on values that sit in one VGPR's lanes.
// Deltas, errors, zigzag, bit widths, field offsets and the packing itself are independent per block.  Phases (barriers between):
//   0  chunk -> LDS; the stream image zeroed
//   1  deltas, transposed to [block][column][8 rows]
//   2  FIRE only, wave 0: the coefficient of every (block, column) -- (1) above
//   3  errors with those coefficients, zigzag, bit widths, the scan of the widths over the columns
//   4  wave 0, uniform: the state machine -- where every packed block and its header fields go; run lengths written
//   5  fields OR-ed into the image (ds_or_b32, two dwords a field), the verbatim tail, the stream header
//   6  image -> HBM in 16-byte pieces
#pragma once

#include "decode_fast.h"
#include "encode_kernel.h"

namespace sprintz {

constexpr uint32_t kEncLatMaxChunkBytes = 16u << 10;
struct EncLatCarve {
    uint32_t o_dl, o_zz, o_coef, o_nbx, o_rb, o_wo, o_img, img_cap, total;
};
inline EncLatCarve enc_lat_carve(uint32_t bound_bytes, uint32_t chunk_len, uint32_t D, uint32_t esz)
{
    EncLatCarve c;
    const uint32_t nb = chunk_len / (8u * D), body = nb * 8u * D * esz;
    auto al = [](uint32_t x) { return (x + 15u) & ~15u; };
    c.o_dl = al(chunk_len * esz + 16u);
    c.o_zz = c.o_dl + al(body + 16u);
    c.o_coef = c.o_zz + al(body + 16u);
    c.o_nbx = c.o_coef + al(nb * D * 4u + 16u);
    c.o_rb = c.o_nbx + al(nb * D * 4u + 16u);
    c.o_wo = c.o_rb + al(nb * 4u + 16u);
    c.o_img = c.o_wo + al(nb * 8u + 16u);
    c.img_cap = al(bound_bytes + 48u);
    c.total = c.o_img + c.img_cap;
    return c;
}

// LOW: the low-dim layout (D <= 4 at 8 bits, <= 2 at 16; sprintz_{delta,xff}_lowdim.cpp:39-400): column-major payload, widths rounded
// only W-1 -> W, untruncated coefficient (32-bit multiply at 16 bits), "<" in the tail test of both codecs.
template <int W, bool FIRE, int DP, bool LOW = false>
__global__ void __launch_bounds__(256) encode_lat_kernel(EncodeArgs a, EncLatCarve cv)
{
    using U = typename Elem<W>::U;
    typedef __attribute__((address_space(3))) uint32_t lds_word;
    constexpr int HB = Elem<W>::HB;
    constexpr int ESZ = W / 8;
    constexpr int LOG2DP = DP == 4 ? 2 : DP == 8 ? 3 : DP == 16 ? 4 : DP == 32 ? 5 : 6;
    constexpr int T = 256 / DP;
    constexpr bool TAIL_LE = FIRE && !LOW;       // "<=" at sprintz_xff_rle.cpp:362, "<" in the other three codecs
    constexpr int RL = DP <= 16 ? 4 : 1;         // lanes that share a column in phase 2 (one per odd row)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ uint32_t info[8];                 // groups, elements consumed, bytes written before the tail

    const uint32_t tid = threadIdx.x;
    const uint64_t chunk = blockIdx.x;
    const int D = a.D;
#ifdef SPRINTZ_LAT_TIMING                                  // experiment builds: phase durations (20 ns ticks, 9 bits each) instead of the return value
    uint64_t stamp[8];
    int nstamp = 0;
#define ENC_STAMP() stamp[nstamp++] = wall_clock64()
#else
#define ENC_STAMP()
#endif
    ENC_STAMP();
    const uint64_t first = chunk * (uint64_t)a.chunk_len;
    const uint32_t n = (uint32_t)((a.total_len - first < a.chunk_len) ? (a.total_len - first) : a.chunk_len);
    const uint32_t blk = 8u * (uint32_t)D;
    const uint32_t NB = n / blk;                 // whole blocks this chunk holds
    const uint32_t hdr_bytes = (2u * (uint32_t)D * HB + 7u) >> 3;
    U* const raw = (U*)smem;
    U* const dl = (U*)(smem + cv.o_dl);
    U* const zz = (U*)(smem + cv.o_zz);
    int32_t* const coefs = (int32_t*)(smem + cv.o_coef);       // (the low-dim coefficient is counter >> 1: up to 31 bits at 16-bit data)
    uint32_t* const nbx = (uint32_t*)(smem + cv.o_nbx);
    uint32_t* const rbits = (uint32_t*)(smem + cv.o_rb);
    uint2* const wofs = (uint2*)(smem + cv.o_wo);
    uint8_t* const img = smem + cv.o_img;
    const uint32_t img_a = lds_addr(img);
    uint8_t* const gdst = a.slots + chunk * a.slot_stride;
    const int lane_d = (int)(tid & (uint32_t)(DP - 1));
    const bool col_ok = lane_d < D;

    // ---- 0: the chunk into LDS (the <= 15 bytes read past its end are inside SPRINTZ_MI355X_READ_SLACK); a clean image
    {
        const uint4* g = (const uint4*)((const U*)a.src + first);
        const uint32_t n16 = (n * ESZ + 15u) >> 4;
        for (uint32_t i = tid; i < n16; i += 256u) ((uint4*)raw)[i] = g[i];
        for (uint32_t i = tid; i < (cv.img_cap >> 4); i += 256u) ((uint4*)img)[i] = make_uint4(0, 0, 0, 0);
        for (uint32_t i = tid; i < NB; i += 256u) wofs[i] = make_uint2(0xffffffffu, 0u);
    }
    __syncthreads();
    ENC_STAMP();

    // ---- 1: deltas (:197-205), [block][column][row]
    for (uint32_t b = tid >> LOG2DP; b < NB; b += (uint32_t)T) {
        if (col_ok) {
            const U* const x = raw + (size_t)b * blk + (uint32_t)lane_d;
            uint32_t prev = b == 0 ? 0u : (uint32_t)x[-(int)D];
            U* const q = dl + ((size_t)b * (uint32_t)D + (uint32_t)lane_d) * 8u;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint32_t v = x[(uint32_t)i * (uint32_t)D];
                q[i] = (U)(v - prev);
                prev = v;
            }
        }
    }
    __syncthreads();
    ENC_STAMP();

    // ---- 2: the coefficient chain (:217, :240-241, :273-275): wave 0; RL lanes a column, one odd row each
    if constexpr (FIRE) {
        if (tid < 64u) {
            const uint32_t d = RL == 4 ? tid >> 2 : tid, k = RL == 4 ? tid & 3u : 0u;
            const uint32_t dd = d < (uint32_t)D ? d : 0u;          // (a lane past the last column runs column 0 along)
            int ctr = 0;
            if constexpr (RL == 4) {
                // rows 2k and 2k+1 of (block, column): one word at 16 bits, one half-word at 8
                auto pair_at = [&](uint32_t b) -> uint32_t {
                    if constexpr (W == 16) return ((const uint32_t*)dl)[((size_t)b * (uint32_t)D + dd) * 4u + k];
                    else return ((const uint16_t*)dl)[((size_t)b * (uint32_t)D + dd) * 4u + k];
                };
                uint32_t cur = NB ? pair_at(0) : 0u;
                for (uint32_t b = 0; b < NB; b++) {
                    const uint32_t nxt = pair_at(b + 1u < NB ? b + 1u : b);
                    const int coef = fire_coef<W, LOW>(ctr);
                    coefs[(size_t)b * (uint32_t)D + dd] = coef;                  // (four lanes, one value, one place)
                    const int lo = __builtin_amdgcn_sbfe((int)cur, 0, W), hi = __builtin_amdgcn_sbfe((int)cur, W, W);
                    const int pred = LOW ? fire_predict<W, true>(lo, coef) : __builtin_amdgcn_sbfe(mad24(lo, coef, 0), W, W);
                    const int err = __builtin_amdgcn_sbfe(hi - pred, 0, W);
                    uint32_t g = (uint32_t)mad24(sign_of(err), lo, 0);
                    g += dpp<DPP_QUAD_PERM(1, 0, 3, 2)>(0, g);
                    g += dpp<DPP_QUAD_PERM(2, 3, 0, 1)>(0, g);
                    ctr = wrap_counter<W>(ctr + __builtin_amdgcn_sbfe((int)g, 2, W - 2));
                    cur = nxt;
                }
            } else {
                for (uint32_t b = 0; b < NB; b++) {
                    const U* const q = dl + ((size_t)b * (uint32_t)D + dd) * 8u;
                    const int coef = fire_coef<W, LOW>(ctr);
                    coefs[(size_t)b * (uint32_t)D + dd] = coef;
                    int grad = 0;
#pragma unroll
                    for (int i = 1; i < 8; i += 2) {
                        const int lo = sext<W>((int)q[i - 1]), hi = sext<W>((int)q[i]);
                        const int pred = LOW ? fire_predict<W, true>(lo, coef) : __builtin_amdgcn_sbfe(mad24(lo, coef, 0), W, W);
                        const int err = __builtin_amdgcn_sbfe(hi - pred, 0, W);
                        grad = mad24(sign_of(err), lo, grad);
                    }
                    ctr = wrap_counter<W>(ctr + __builtin_amdgcn_sbfe(grad, 2, W - 2));
                }
            }
        }
        __syncthreads();
    }
    ENC_STAMP();

    // ---- 3: errors, zigzag, widths (:225-298), the scan of the widths over the columns
    for (uint32_t b = tid >> LOG2DP; b < NB; b += (uint32_t)T) {
        const size_t at = ((size_t)b * (uint32_t)D + (uint32_t)(col_ok ? lane_d : 0)) * 8u;
        const U* const q = dl + at;
        int pd = b == 0 ? 0 : sext<W>((int)q[-(int)(8 * D) + 7]);              // the last delta of the block before
        const int coef = FIRE ? (int)coefs[(size_t)b * (uint32_t)D + (uint32_t)(col_ok ? lane_d : 0)] : 0;
        uint32_t mask = 0;
        uint32_t z[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int delta = sext<W>((int)q[i]);
            const int pred = FIRE ? (LOW ? fire_predict<W, true>(pd, coef) : __builtin_amdgcn_sbfe(mad24(pd, coef, 0), W, W)) : 0;
            z[i] = zigzag<W>(sext<W>(delta - pred));
            mask |= z[i];
            pd = delta;
        }
        const uint32_t nb = col_ok ? nbits_of<W, LOW>(mask) : 0u;
        uint32_t total;
        const uint32_t excl = group_scan<DP>(nb, lane_d, total);
        if (col_ok) {
            U* const o = zz + at;
#pragma unroll
            for (int i = 0; i < 8; i++) o[i] = (U)z[i];
            nbx[(size_t)b * (uint32_t)D + (uint32_t)lane_d] = nb | (excl << 8);
        }
        if (lane_d == 0) rbits[b] = total;
    }
    __syncthreads();
    ENC_STAMP();

    // ---- 4: the RLE / group state machine (:350-456) -- as SCANS.  Run as the reference writes it (one block at a
    // time, ~45 scalar instructions and ten branches each) it was the longest phase of a chunk: >= 10 us for 80 blocks, 150 for a
    // univariate chunk's 1 280.  But its decisions are functions of positions only:
    //   * a SLOT is a block that is not all zero, or a maximal run of all-zero blocks (the 32 767-block cap cannot bind: a chunk has
    //     at most 2 048 blocks); slot k goes to group k / 2, and a group's header sits in front of its first slot;
    //   * the walk STOPS after the first block b that is (i) all zero with b + 1 past the run limit -- "<=" against
    //     last_full_group_start in the general FIRE codec, "<" in the other three (:362) -- or (ii) packed, second slot of its
    //     group, with b + 1 past last_full_group_start (:450); a run closed by a packed block opens the next group WITHOUT that
    //     test (:430-450), which is just the slot arithmetic; a run that stops the stream in a group's first slot leaves one 0x00
    //     for the second (:386-391);
    //   * a run's varint sits where the run ENDS (its length is the distance to the last packed block before it).
    // So: wave 0, lane l owns blocks [l P, (l + 1) P), P = ceil(blocks / 64); slot numbers, the stop block, byte offsets and
    // header positions are one wave scan each with a short pass over the lane's own blocks in between.
    if (tid < 64u) {
        const int64_t limit = (int64_t)n - 2 * (int64_t)blk;        // last_full_group_start (:158), in elements
        const int32_t lim_le = limit >= 0 ? (int32_t)(limit / (int64_t)blk) : -1;              // pos_in <= limit  <=>  b <= lim_le
        const int32_t lim_lt = limit > 0 ? (int32_t)((limit - 1) / (int64_t)blk) : -1;         // pos_in <  limit  <=>  b <= lim_lt
        const int32_t lim_run = TAIL_LE ? lim_le : lim_lt;
        const uint32_t base_wl = a.write_size ? 8u : 0u;
        const uint32_t slot_bits = (uint32_t)D * HB;
        if (!(n >= 128u && limit >= 0)) {                            // :116 and the loop guard :160: header + verbatim samples
            if (tid == 0) { info[0] = 0; info[1] = 0; info[2] = base_wl; }
        } else {
            const uint32_t P = (NB + 63u) >> 6;
            const uint32_t i0 = tid * P < NB ? tid * P : NB, i1 = i0 + P < NB ? i0 + P : NB;
            auto payload = [&](uint32_t rb) -> uint32_t { return LOW ? rb : ((rb + 7u) >> 3) << 3; };
            const bool first_prev = i0 == 0 ? true : rbits[i0 - 1] != 0;      // (block 0 starts a slot whatever it is)
            // pass 1: slot starts and the last packed block of the lane's piece
            uint32_t starts = 0;
            int lnz = -1;
            {
                bool prev = first_prev;
                for (uint32_t bb = i0; bb < i1; bb++) {
                    const bool nz = rbits[bb] != 0;
                    starts += (nz || prev) ? 1u : 0u;
                    lnz = nz ? (int)bb : lnz;
                    prev = nz;
                }
            }
            uint32_t starts_total;
            const uint32_t starts_before = group_scan<64>(starts, (int)tid, starts_total);
            int lnz_before = lnz;                                    // exclusive max-scan: the last packed block before the piece
            {
                int incl = lnz;
                for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if ((int)tid >= off) incl = incl > t ? incl : t; }
                lnz_before = __shfl_up(incl, 1, 64);
                if (tid == 0) lnz_before = -1;
            }
            // pass 2: the block the walk stops after
            uint32_t cand = 0xffffffffu;
            {
                bool prev = first_prev;
                uint32_t sid = starts_before;                        // slots started so far
                for (uint32_t bb = i0; bb < i1; bb++) {
                    const bool nz = rbits[bb] != 0;
                    sid += (nz || prev) ? 1u : 0u;
                    const bool second = ((sid - 1u) & 1u) != 0;
                    const bool stop = nz ? (second && (int32_t)(bb + 1u) > lim_le) : ((int32_t)(bb + 1u) > lim_run);
                    cand = (stop && bb < cand) ? bb : cand;
                    prev = nz;
                }
            }
            for (int off = 32; off > 0; off >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)cand, off, 64); cand = t < cand ? t : cand; }
            const uint32_t stop_b = cand < NB ? cand : NB - 1u;      // (it always exists: the last two blocks trip one of the tests)
            // pass 3: bytes of the lane's piece -- group headers in front of even slots, payloads, run lengths where runs end
            uint32_t bytes = 0;
            {
                bool prev = first_prev;
                uint32_t sid = starts_before;
                int last = lnz_before;
                for (uint32_t bb = i0; bb < i1 && bb <= stop_b; bb++) {
                    const uint32_t rb = rbits[bb];
                    const bool nz = rb != 0, st = nz || prev;
                    sid += st ? 1u : 0u;
                    if (st && ((sid - 1u) & 1u) == 0) bytes += hdr_bytes;
                    if (nz) { bytes += payload(rb); last = (int)bb; }
                    else if (bb == stop_b || rbits[bb + 1u] != 0) bytes += ((int)bb - last) > 127 ? 2u : 1u;
                    prev = nz;
                }
            }
            uint32_t bytes_total;
            const uint32_t bytes_before = group_scan<64>(bytes, (int)tid, bytes_total);
            // pass 4a: where the last group header of the piece sits; carried to the lanes behind by a max-scan (positions only grow)
            uint32_t hp = 0;
            {
                bool prev = first_prev;
                uint32_t sid = starts_before, off = base_wl + bytes_before;
                int last = lnz_before;
                for (uint32_t bb = i0; bb < i1 && bb <= stop_b; bb++) {
                    const uint32_t rb = rbits[bb];
                    const bool nz = rb != 0, st = nz || prev;
                    sid += st ? 1u : 0u;
                    if (st && ((sid - 1u) & 1u) == 0) { hp = off; off += hdr_bytes; }
                    if (nz) { off += payload(rb); last = (int)bb; }
                    else if (bb == stop_b || rbits[bb + 1u] != 0) off += ((int)bb - last) > 127 ? 2u : 1u;
                    prev = nz;
                }
            }
            uint32_t hp_before;
            {
                uint32_t incl = hp;
                for (int off = 1; off < 64; off <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, off, 64); if ((int)tid >= off) incl = incl > t ? incl : t; }
                hp_before = (uint32_t)__shfl_up((int)incl, 1, 64);
                if (tid == 0) hp_before = 0;
            }
            // pass 4b: the packed blocks' places, the run lengths themselves, the totals
            {
                bool prev = first_prev;
                uint32_t sid = starts_before, off = base_wl + bytes_before, cur_hp = hp_before;
                int last = lnz_before;
                for (uint32_t bb = i0; bb < i1 && bb <= stop_b; bb++) {
                    const uint32_t rb = rbits[bb];
                    const bool nz = rb != 0, st = nz || prev;
                    sid += st ? 1u : 0u;
                    const uint32_t par = (sid - 1u) & 1u;
                    if (st && par == 0) { cur_hp = off; off += hdr_bytes; }
                    uint32_t pad = 0;
                    if (nz) {
                        wofs[bb] = make_uint2(off, cur_hp * 8u + par * slot_bits);
                        off += payload(rb);
                        last = (int)bb;
                    } else if (bb == stop_b || rbits[bb + 1u] != 0) {
                        const uint32_t r = (uint32_t)((int)bb - last);                 // :377-384
                        img[off] = (uint8_t)((r & 0x7fu) | (r > 0x7fu ? 0x80u : 0u));
                        if (r > 0x7fu) img[off + 1] = (uint8_t)(r >> 7);
                        off += r > 0x7fu ? 2u : 1u;
                        if (bb == stop_b) pad = 1u - par;            // one 0x00 per slot the group still has (:386-391)
                    }
                    if (bb == stop_b) { info[0] = ((sid - 1u) >> 1) + 1u; info[1] = (stop_b + 1u) * blk; info[2] = base_wl + bytes_total + pad; }
                    prev = nz;
                }
            }
        }
    }
    __syncthreads();
    ENC_STAMP();
    const uint32_t ngroups = info[0], pos_in = info[1], wl = info[2];
    const uint32_t remaining = n - pos_in;

    // ---- 5: the fields into the image (:483-524), the verbatim tail (:553), the stream header (format.h:36-45)
    auto or_bits = [&](uint32_t bp, uint32_t v, uint32_t nb) {      // the low nb (<= 16) bits of v at image bit position bp
        if (nb == 0) return;
        const uint64_t x = (uint64_t)v << (bp & 31u);
        lds_word* q = (lds_word*)(uintptr_t)(img_a + ((bp >> 3) & ~3u));
        __hip_atomic_fetch_or(q, (uint32_t)x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((uint32_t)(x >> 32)) __hip_atomic_fetch_or(q + 1, (uint32_t)(x >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    for (uint32_t b = tid >> LOG2DP; b < NB; b += (uint32_t)T) {
        const uint2 wo = wofs[b];
        if (wo.x == 0xffffffffu || !col_ok) continue;               // a block inside a run, or past the last one coded
        const uint32_t e = nbx[(size_t)b * (uint32_t)D + (uint32_t)lane_d];
        const uint32_t nb = e & 0xffu, excl = e >> 8;
        const uint32_t row_bits = ((rbits[b] + 7u) >> 3) << 3;
        or_bits(wo.y + (uint32_t)lane_d * HB, nb == (uint32_t)W ? (uint32_t)(W - 1) : nb, HB);     // :296
        const U* const zq = zz + ((size_t)b * (uint32_t)D + (uint32_t)lane_d) * 8u;
        uint32_t bp = LOW ? (wo.x + excl) * 8u : wo.x * 8u + excl;          // low-dim: this column's bytes behind the columns before it
#pragma unroll
        for (int i = 0; i < 8; i++) {
            or_bits(bp, (uint32_t)zq[i], nb);
            bp += LOW ? nb : row_bits;
        }
    }
    {
        const uint8_t* const tp = (const uint8_t*)(raw + pos_in);
        for (uint32_t j = tid; j < remaining * ESZ; j += 256u) img[wl + j] = tp[j];
    }
    __syncthreads();
    if (tid == 0 && a.write_size) {                                  // (after the ORs: the image's first 8 bytes are plain stores)
        ((uint32_t*)img)[0] = ngroups;
        ((uint32_t*)img)[1] = (remaining & 0xffffu) | ((uint32_t)D << 16);
    }
    __syncthreads();
    ENC_STAMP();

    // ---- 6: the stream leaves in 16-byte pieces (the slot is 16-byte aligned and holds the bound)
    const uint32_t total_bytes = wl + remaining * ESZ;
    for (uint32_t i = tid; i < ((total_bytes + 15u) >> 4); i += 256u) ((uint4*)gdst)[i] = ((const uint4*)img)[i];
    ENC_STAMP();
    if (a.host_flag) { __threadfence_system(); __syncthreads(); }      // a single call on mapped host memory: every lane's stores, then the ticket
    if (tid == 0) {
        a.sizes[chunk] = total_bytes;
        if (a.rets) a.rets[chunk] = (int64_t)(total_bytes / ESZ);
        if (a.host_flag) {
            __threadfence_system();
            __hip_atomic_store(a.host_flag, a.host_ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
#ifdef SPRINTZ_LAT_TIMING
        if (a.rets) {
            uint64_t r = 0;
            for (int k = 0; k < 7; k++) { const uint64_t dt = (stamp[k + 1] - stamp[k]) >> 1; r |= (dt < 511 ? dt : 511) << (9 * k); }
            a.rets[chunk] = (int64_t)r;
        }
#endif
    }
#undef ENC_STAMP
}

}  // namespace sprintz

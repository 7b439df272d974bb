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

    // ---- 4: the RLE / group state machine (:350-456), scalar: the blocks' row widths sit in the lanes of one
    // register, 64 blocks at a time; where a packed block goes comes back the same way
    if (tid < 64u) {
        const int64_t limit = (int64_t)n - 2 * (int64_t)blk;        // last_full_group_start (:158)
        uint32_t wl = a.write_size ? 8u : 0u;                        // bytes of stream so far
        uint32_t ngroups = 0, run = 0, hdr_pos = 0, slot = 0, b = 0;
        bool active = n >= 128u && limit >= 0;                       // :116 and the loop guard :160
        auto start_group = [&]() { ngroups++; hdr_pos = wl; wl += hdr_bytes; slot = 0; };
        auto put_run = [&](uint32_t r) {                             // :377-384
            if (tid == 0) {
                img[wl] = (uint8_t)((r & 0x7fu) | (r > 0x7fu ? 0x80u : 0u));
                if (r > 0x7fu) img[wl + 1] = (uint8_t)(r >> 7);
            }
            wl += r > 0x7fu ? 2u : 1u;
        };
        if (active) start_group();
        for (uint32_t base = 0; active && base < NB; base += 64u) {
            const int widths = base + tid < NB ? (int)rbits[base + tid] : 0;
            int wo_x = -1, wo_y = 0;
            while (active && b < base + 64u) {
                const uint32_t rb = (uint32_t)__builtin_amdgcn_readlane(widths, (int)(b - base));
                if (rb == 0 && run < 0x7fffu) {
                    run++;
                    b++;
                    const int64_t pos_in = (int64_t)b * blk;
                    if (TAIL_LE ? (pos_in <= limit) : (pos_in < limit)) continue;
                    slot++;
                    put_run(run);
                    wl += 2u - slot;                                 // one 0x00 per slot the group still has (:386-391)
                    run = 0;
                    active = false;
                    break;
                }
                if (run > 0) {                                       // a block that is not all zero closes the run
                    slot++;
                    put_run(run);
                    run = 0;
                    if (slot == 2) start_group();                    // :430-450 (no look at the limit here)
                }
                const bool me = tid == b - base;                     // (a lane write: compare + select on scalars)
                wo_x = me ? (int)wl : wo_x;
                wo_y = me ? (int)(hdr_pos * 8u + slot * (uint32_t)D * HB) : wo_y;
                wl += LOW ? rb : ((rb + 7u) >> 3) << 3;              // 8 rows of ceil(rb / 8) bytes; low-dim: a column's 8 values are nbits bytes
                b++;
                slot++;
                if (slot == 2) {
                    if ((int64_t)b * blk <= limit) start_group();
                    else active = false;
                }
            }
            if (base + tid < NB) wofs[base + tid] = make_uint2((uint32_t)wo_x, (uint32_t)wo_y);
        }
        if (tid == 0) { info[0] = ngroups; info[1] = b * blk; info[2] = wl; }
    }
    __syncthreads();
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

    // ---- 6: the stream leaves in 16-byte pieces (the slot is 16-byte aligned and holds the bound)
    const uint32_t total_bytes = wl + remaining * ESZ;
    for (uint32_t i = tid; i < ((total_bytes + 15u) >> 4); i += 256u) ((uint4*)gdst)[i] = ((const uint4*)img)[i];
    if (tid == 0) {
        a.sizes[chunk] = total_bytes;
        if (a.rets) a.rets[chunk] = (int64_t)(total_bytes / ESZ);
    }
}

}  // namespace sprintz

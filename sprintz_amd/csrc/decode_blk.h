// decode_blk.h -- the BLOCK-PARALLEL decoder of the DELTA codec, general (row-major) payload layout, for large batches:
// the inverse of encode_blk.h, same task shape.  Decodes every stream decode_kernel.h decodes
// (decompress_rowmajor_delta_rle, sprintz_delta_rle.cpp:418-772) to the same samples and return values.
//
// Delta decoding is a prefix sum down every column, and the only serial thing in the FORMAT is where a
// group starts -- the sum of all header fields in front of it.  So, per chunk:
//   0  the stream into an LDS image, 16 bytes a thread (the only global reads of the launch);
//   1  the WALK: 16 lanes a chunk read a group header a dword each, reduce the two slots' widths (DPP), step over payloads and
//      run lengths, and leave one descriptor per OUTPUT block: where its payload and its header fields are, or "all zero" (a run);
//   2  a TASK = (block of 8 rows, 16-byte piece of the row) = 16 uint8 / 8 uint16 columns: widths from the header (48 / 32 bits),
//      the piece's bit offset inside the row from the other pieces' sums (LDS), then per row and per dword of output ONE unaligned
//      32-bit window from the image (ds_read2_b32 + v_alignbit) holding its 4 (2) fields; zigzag^-1 fused with the field fetch
//      (v_bfe_u32 / v_bfe_i32 / v_xor), the running sum down the block's rows kept PACKED (an SDWA add per sample writes its byte of
//      the row's dword: no pack step);
//   3  the blocks' totals (row 7 of each task, 16 bytes) are scanned down the chunk's blocks per piece: lane = (piece, dword);
//   4  every task adds its carry to its 8 row pieces (carry-isolated byte adds / v_pk_add_u16) and stores them, 16 bytes a row.
// A damaged stream never moves a cursor past its end (the walk checks every step against the stream's length; tasks only follow
// descriptors the walk has validated) and decodes to nothing: SPRINTZ_E_CORRUPT.
#pragma once

#include "decode_fast.h"
#include "group_ops.h"

namespace sprintz {

struct BlkDecGeom {
    uint32_t P, NBC, T, CPW;
    uint32_t img_cap;                                    // bytes of one chunk's stream image (multiple of 16)
    uint32_t o_desc, o_psum, o_csum, o_info, total;      // LDS carve (bytes)
    uint32_t invT, invP;                                 // ceil(2^16 / T), ceil(2^16 / P): n / d == (n * inv) >> 16 for n < 256, d <= 256
    uint32_t ok;
};

inline BlkDecGeom blk_dec_geom(uint32_t esz, uint32_t chunk_len, uint32_t D, uint32_t bound_bytes)
{
    BlkDecGeom g{};
    const uint32_t rowbytes = D * esz, hb = esz == 1 ? 3u : 4u;
    if (rowbytes % 16u || ((uint64_t)chunk_len * esz) % 16u || chunk_len < 32u * D) return g;      // (>= 4 blocks: the scan's lanes are tasks)
    if (2u * D > 16u * (hb == 3u ? 10u : 8u)) return g;  // a group header's 2 D fields over the walk's 16 lanes: 10 x 3 / 8 x 4 bits each (80 / 64 columns)
    g.P = rowbytes / 16u;
    g.NBC = chunk_len / (8u * D);
    g.T = g.NBC * g.P;
    if (g.T > 256u || g.NBC >= 32767u) return g;
    g.img_cap = (bound_bytes + 64u + 15u) & ~15u;        // + the start's phase in its 16-byte piece, + windows that look past the last byte
    auto al = [](uint32_t x) { return (x + 15u) & ~15u; };
    uint32_t cpw = 256u / g.T;
    if (cpw > 16u) cpw = 16u;                            // (four wavefronts walk four chunks each)
    for (; cpw >= 1; cpw--) {
        g.CPW = cpw;
        g.o_desc = cpw * g.img_cap;
        g.o_psum = g.o_desc + al(cpw * g.NBC * 8u);
        g.o_csum = g.o_psum + al(cpw * g.T * 2u);
        g.o_info = g.o_csum + cpw * g.T * 16u;
        g.total = g.o_info + cpw * 16u;
        if (g.total <= 64u * 1024u) break;
    }
    g.invT = (65536u + g.T - 1u) / g.T;
    g.invP = (65536u + g.P - 1u) / g.P;
    g.ok = cpw >= 1 ? 1u : 0u;
    return g;
}

// bytewise / halfword-wise a + b
template <int W> __device__ __forceinline__ uint32_t lanes_add(uint32_t a, uint32_t b)
{
    if constexpr (W == 8) {
        constexpr uint32_t H = 0x80808080u;
        return ((a & ~H) + (b & ~H)) ^ ((a ^ b) & H);
    } else {
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(uint32_t, __builtin_bit_cast(us2, a) + __builtin_bit_cast(us2, b));
    }
}

// dst.field<F> = src.field<F> + (low field of e); the other fields of dst: zero (F == 0, the first write of a row's dword) or kept
template <int W, int F> __device__ __forceinline__ void field_acc(uint32_t& dst, uint32_t src, uint32_t e)
{
    if constexpr (W == 8) {
        if constexpr (F == 0) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0" : "=v"(dst) : "v"(src), "v"(e));
        else if constexpr (F == 1) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:BYTE_0" : "+v"(dst) : "v"(src), "v"(e));
        else if constexpr (F == 2) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2 src1_sel:BYTE_0" : "+v"(dst) : "v"(src), "v"(e));
        else asm("v_add_u32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:BYTE_0" : "+v"(dst) : "v"(src), "v"(e));
    } else {
        if constexpr (F == 0) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0" : "=v"(dst) : "v"(src), "v"(e));
        else asm("v_add_u32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_0" : "+v"(dst) : "v"(src), "v"(e));
    }
}

constexpr uint32_t kBlkZero = 0xffffffffu;               // descriptor of a block inside a run: every error zero

template <int W>
__global__ void __launch_bounds__(256) decode_blk_kernel(DecodeArgs a, BlkDecGeom g)
{
    constexpr int HB = Elem<W>::HB;
    constexpr int ESZ = W / 8;
    constexpr int CPP = 16 / ESZ;                // columns per piece
    constexpr int FPD = 4 / ESZ;                 // fields per dword
    constexpr uint32_t FM = (1u << HB) - 1u;
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const uint32_t tid = threadIdx.x;
    const uint32_t ci = (tid * g.invT) >> 16, k = tid - ci * g.T;
    const uint32_t b = (k * g.invP) >> 16, p = k - b * g.P;
    const bool in_wg = ci < g.CPW;
    const uint64_t chunk = (uint64_t)blockIdx.x * g.CPW + ci;
    const bool exists = in_wg && chunk < a.nchunks;
    const uint32_t D = (uint32_t)a.D, blk = 8u * D, rowbytes = D * ESZ;
    const uint32_t cix = in_wg ? ci : 0u;
    uint8_t* const img = smem + cix * g.img_cap;
    const uint32_t img_a = lds_addr(img);
    uint2* const desc = (uint2*)(smem + g.o_desc) + cix * g.NBC;
    uint16_t* const psum = (uint16_t*)(smem + g.o_psum) + cix * g.T;
    v4* const csum = (v4*)(smem + g.o_csum) + cix * g.T;
    uint32_t* const info = (uint32_t*)(smem + g.o_info) + cix * 4u;

    // ---- 0: the stream into its image: 16-byte pieces from the piece that holds its first byte (the container's read slack covers the last)
    const uint64_t off = exists ? a.offsets[chunk] : 0ull;
    const uint64_t slen64 = exists ? a.offsets[chunk + 1] - off : 0ull;
    const uint32_t phase = (uint32_t)(((uintptr_t)a.comp + off) & 15u);       // of the stream's first byte in its 16-byte piece of MEMORY
    const bool fits = slen64 + phase + 32u <= (uint64_t)g.img_cap;            // a longer stream is no stream of this shape
    const uint32_t slen = fits ? (uint32_t)slen64 : 0u;
    if (exists && fits) {
        const v4* const src = (const v4*)(a.comp + (off - phase));
        const uint32_t n16 = (phase + slen + 15u) >> 4;                       // (the last piece reaches <= 15 bytes past the stream: inside the container's read slack)
        for (uint32_t i = k; i < n16; i += g.T) ((v4*)img)[i] = __builtin_nontemporal_load(src + i);
        if (k == 0) ((v4*)img)[n16] = v4{0u, 0u, 0u, 0u};                     // windows look up to 11 bytes past the last field: defined bytes
    }
    __syncthreads();

    // ---- 1: the walk: wavefront w takes the chunks [4 w, 4 w + 4) of the workgroup, 16 lanes each
    {
        const uint32_t wave = tid >> 6, lane = tid & 63u;
        if (wave * 4u < g.CPW) {                                             // (uniform per wavefront)
            const uint32_t wc = wave * 4u + (lane >> 4), lg = lane & 15u;
            const uint32_t wcx = wc < g.CPW ? wc : 0u;
            const uint64_t wchunk = (uint64_t)blockIdx.x * g.CPW + wc;
            const bool wlive = wc < g.CPW && wchunk < a.nchunks;
            const uint64_t woff = wlive ? a.offsets[wchunk] : 0ull;
            const uint64_t wslen64 = wlive ? a.offsets[wchunk + 1] - woff : 0ull;
            const uint32_t wphase = (uint32_t)(((uintptr_t)a.comp + woff) & 15u);
            const bool wfits = wslen64 + wphase + 32u <= (uint64_t)g.img_cap;
            const uint32_t wslen = wfits ? (uint32_t)wslen64 : 0u;
            const uint32_t wimg = lds_addr(smem + wcx * g.img_cap) + wphase;  // LDS byte address of the stream's first byte
            uint2* const wdesc = (uint2*)(smem + g.o_desc) + wcx * g.NBC;
            uint32_t* const winfo = (uint32_t*)(smem + g.o_info) + wcx * 4u;
            const uint32_t hdr_bytes = (2u * D * HB + 7u) >> 3;

            bool corrupt = !wlive || !wfits || wslen < 8u;
            uint32_t groups_left = 0, remaining = 0, pos = 8u, bout = 0;
            if (!corrupt) {
                const uint32_t w0 = lds_rd32(wimg), w1 = lds_rd32(wimg + 4u);
                groups_left = w0;
                remaining = w1 & 0xffffu;
                // a damaged header must not make the loop spin: every group of a valid stream holds at least one non-empty slot, except the last
                corrupt = (w1 >> 16) != D || groups_left > a.chunk_len / blk + 2u;
            }
            if (corrupt) groups_left = 0;
            // A group header is 2 D fields of HB bits.  Lane lg of the chunk's 16 takes FPL of them -- 10 x 3 / 8 x 4 bits: one 32-bit window at a bit
            // address -- the first n0 of which belong to slot 0; both slots' width sums are population counts of the window's bit planes
            // (width = field, + 1 where the field is all ones: W - 1 means W, :747-749), then four DPP adds over the 16 lanes.
            constexpr uint32_t FPL = HB == 3 ? 10u : 8u;
            const uint32_t f_lo = FPL * lg;
            const uint32_t fv = 2u * D > f_lo ? (2u * D - f_lo < FPL ? 2u * D - f_lo : FPL) : 0u;      // fields of this lane that exist
            const uint32_t n0 = D > f_lo ? (D - f_lo < fv ? D - f_lo : fv) : 0u;                        // ... that belong to slot 0
            const uint32_t maskv = fv * HB >= 32u ? 0xffffffffu : (1u << (fv * HB)) - 1u;
            const uint32_t mask0 = n0 * HB >= 32u ? 0xffffffffu : (1u << (n0 * HB)) - 1u;
            const uint32_t mask1 = maskv & ~mask0;
            constexpr uint32_t PL = HB == 3 ? 0x09249249u : 0x11111111u;                                // bit 0 of every field
            auto width_sum = [&](uint32_t y) -> uint32_t {
                const uint32_t p0 = y & PL, p1 = (y >> 1) & PL, p2 = (y >> 2) & PL;
                if constexpr (HB == 3) {
                    return (uint32_t)__builtin_popcount(p0) + (uint32_t)__builtin_popcount(p0 & p1 & p2) + 2u * (uint32_t)__builtin_popcount(p1) + 4u * (uint32_t)__builtin_popcount(p2);
                } else {
                    const uint32_t p3 = (y >> 3) & PL;
                    return (uint32_t)__builtin_popcount(p0) + (uint32_t)__builtin_popcount(p0 & p1 & p2 & p3) + 2u * (uint32_t)__builtin_popcount(p1) +
                           4u * (uint32_t)__builtin_popcount(p2) + 8u * (uint32_t)__builtin_popcount(p3);
                }
            };
            const uint32_t slot_bits = D * HB;
            while (__ballot(groups_left != 0u) != 0ull) {
                const bool act = groups_left != 0u;
                // (lanes of a chunk that is done run along on position 0 of their own image: every read stays inside the carve)
                const uint32_t hpos = act ? pos : 0u;
                const bool hdr_ok = hdr_bytes <= wslen - hpos || !act;
                uint32_t y = 0;
                {
                    typedef __attribute__((address_space(3))) const uint32_t lds_cw;
                    const uint32_t A = (wimg + hpos) * 8u + f_lo * HB;
                    lds_cw* q = (lds_cw*)(uintptr_t)((A >> 3) & ~3u);
                    y = __builtin_amdgcn_alignbit(q[1], q[0], A & 31u);
                }
                uint32_t both = width_sum(y & mask0) | (width_sum(y & mask1) << 16);   // (<= 80 x 8 / 64 x 16 a slot: no carry between the halves)
                both += dpp<DPP_QUAD_PERM(1, 0, 3, 2)>(0u, both);
                both += dpp<DPP_QUAD_PERM(2, 3, 0, 1)>(0u, both);
                both += dpp<DPP_ROW_HALF_MIRROR>(0u, both);
                both += dpp<DPP_ROW_MIRROR>(0u, both);
                const uint32_t S0 = both & 0xffffu, S1 = both >> 16;
                const uint32_t pay0 = ((S0 + 7u) >> 3) << 3, pay1 = ((S1 + 7u) >> 3) << 3;
                // the common group: two packed blocks that fit the stream and the chunk
                const bool fast = act && hdr_ok && S0 != 0u && S1 != 0u && hdr_bytes + pay0 + pay1 <= wslen - hpos && bout + 2u <= g.NBC;
                if (fast) {
                    if (lg < 2u) wdesc[bout + lg] = make_uint2(hpos + hdr_bytes + (lg ? pay0 : 0u), hpos * 8u + (lg ? slot_bits : 0u));
                    bout += 2u;
                    pos = hpos + hdr_bytes + pay0 + pay1;
                    groups_left -= 1u;
                }
                if (__ballot(act && !fast) != 0ull) {            // a run, a padding slot, the chunk's end, damage: slot by slot
                    if (act && !fast) {
                        bool bad = !hdr_ok;
                        uint32_t cur = hpos + hdr_bytes;
#pragma unroll
                        for (int slot = 0; slot < 2; slot++) {
                            const uint32_t S = slot ? S1 : S0;
                            if (S == 0u) {                                            // RUN slot: length in blocks, 1 or 2 bytes (:829-833)
                                if (!bad && wslen - cur < 2u) bad = wslen == cur || (lds_rd8(wimg + cur) & 0x80u) != 0u;
                                const uint32_t b0 = bad ? 0u : lds_rd8(wimg + cur);
                                uint32_t len = b0 & 0x7fu;
                                if (b0 & 0x80u) { len |= lds_rd8(wimg + cur + 1u) << 7; cur += 2u; }
                                else cur += 1u;
                                if (!bad && len != 0u) {
                                    if (bout + len > g.NBC) bad = true;
                                    else {
                                        for (uint32_t q = bout + lg; q < bout + len; q += 16u) wdesc[q] = make_uint2(kBlkZero, 0u);
                                        bout += len;
                                    }
                                }
                            } else if (!bad) {                                        // packed block: 8 rows of ceil(S / 8) bytes
                                const uint32_t pay = slot ? pay1 : pay0;
                                if (pay > wslen - cur || bout >= g.NBC) bad = true;
                                else {
                                    if (lg == 0) wdesc[bout] = make_uint2(cur, hpos * 8u + (uint32_t)slot * slot_bits);
                                    bout += 1u;
                                    cur += pay;
                                }
                            }
                        }
                        pos = cur;
                        groups_left -= 1u;
                        if (bad) { corrupt = true; groups_left = 0u; }
                    }
                }
            }
            // the verbatim tail must fit both ways (:1171)
            if (!corrupt && (bout * blk + remaining > a.chunk_len || (uint64_t)remaining * ESZ > (uint64_t)(wslen - pos))) corrupt = true;
            if (lg == 0 && wc < g.CPW) { winfo[0] = corrupt ? 0u : bout; winfo[1] = pos; winfo[2] = corrupt ? 0u : remaining; winfo[3] = corrupt ? 1u : 0u; }
        }
    }
    __syncthreads();
    const uint32_t nbo = info[0], tail_pos = info[1], remaining = info[2];
    const bool corrupt = info[3] != 0u;
    const bool task = exists && b < nbo;
    const uint32_t sbit = (img_a + phase) * 8u;          // LDS BIT address of the stream's first bit (the carve is far below 2^29 bytes)

    // ---- 2: widths, errors, the running sum down the block's rows
    const uint2 de = task ? desc[b] : make_uint2(kBlkZero, 0u);
    const bool packed = task && de.x != kBlkZero;
    uint32_t nb[4][FPD], wsum[4], S = 0;
    {
        uint64_t hv = 0;
        if (packed) {
            const uint32_t hb = sbit + de.y + p * (uint32_t)(CPP * HB);
            typedef __attribute__((address_space(3))) const uint32_t lds_cw;
            lds_cw* q = (lds_cw*)(uintptr_t)((hb >> 3) & ~3u);
            const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], s = hb & 31u;
            hv = (uint64_t)__builtin_amdgcn_alignbit(d1, d0, s) | ((uint64_t)__builtin_amdgcn_alignbit(d2, d1, s) << 32);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            wsum[j] = 0;
#pragma unroll
            for (int f = 0; f < FPD; f++) {
                const uint32_t fv = (uint32_t)(hv >> (HB * (j * FPD + f))) & FM;
                nb[j][f] = fv + ((fv + 1u) >> HB);
                wsum[j] += nb[j][f];
            }
            S += wsum[j];
        }
    }
    if (task) psum[k] = (uint16_t)S;
    __syncthreads();
    uint32_t Bp = 0, tot = 0;
    if (packed) {
        for (uint32_t q = 0; q < g.P; q++) {
            const uint32_t s = psum[b * g.P + q];
            Bp += q < p ? s : 0u;
            tot += s;
        }
    }
    uint32_t acc[8][4];
#pragma unroll
    for (int r = 0; r < 8; r++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[r][j] = 0;
    if (packed) {
        const uint32_t row_bits = ((tot + 7u) >> 3) << 3;
        // per field: shift, shift + 1, width of the magnitude, width of the sign (zigzag^-1 fused with the fetch: err = bfe_u(t, s + 1, n - 1) ^ bfe_i(t, s, 1))
        uint32_t fs[4][FPD], fs1[4][FPD], wm[4][FPD], w1[4][FPD], cbit[4];
        uint32_t c = sbit + de.x * 8u + Bp;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            cbit[j] = c;
            uint32_t s = 0;
#pragma unroll
            for (int f = 0; f < FPD; f++) {
                fs[j][f] = s;
                fs1[j][f] = s + 1u;
                w1[j][f] = nb[j][f] != 0u ? 1u : 0u;
                wm[j][f] = nb[j][f] - w1[j][f];
                s += nb[j][f];
            }
            c += wsum[j];
        }
        typedef __attribute__((address_space(3))) const uint32_t lds_cw;
        uint32_t rb = 0;
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t bp = cbit[j] + rb;
                lds_cw* q = (lds_cw*)(uintptr_t)((bp >> 3) & ~3u);
                const uint32_t t = __builtin_amdgcn_alignbit(q[1], q[0], bp & 31u);   // the unit's <= 32 bits
                uint32_t row = 0;
#pragma unroll
                for (int f = 0; f < FPD; f++) {
                    const uint32_t e = __builtin_amdgcn_ubfe(t, fs1[j][f], wm[j][f]) ^ (uint32_t)__builtin_amdgcn_sbfe((int)t, fs[j][f], w1[j][f]);
                    const uint32_t prev = r ? acc[r - 1][j] : 0u;
                    if (f == 0) field_acc<W, 0>(row, prev, e);
                    else if (f == 1) field_acc<W, 1>(row, prev, e);
                    else if (f == 2) field_acc<W, (FPD > 2 ? 2 : 1)>(row, prev, e);
                    else field_acc<W, (FPD > 2 ? 3 : 1)>(row, prev, e);
                }
                acc[r][j] = row;
            }
            rb += row_bits;
        }
    }
    if (task) csum[k] = v4{acc[7][0], acc[7][1], acc[7][2], acc[7][3]};
    __syncthreads();

    // ---- 3: the blocks' totals scanned down the chunk, lane = (piece, dword): exclusive, in place
    if (exists && k < 4u * g.P && nbo != 0u) {
        const uint32_t sp = k >> 2, sj = k & 3u;
        uint32_t* const col = (uint32_t*)csum + sp * 4u + sj;                 // block q's word at col[q * P * 4]
        uint32_t run = 0;
        for (uint32_t q = 0; q < nbo; q++) {
            const uint32_t v = col[(size_t)q * g.P * 4u];
            col[(size_t)q * g.P * 4u] = run;
            run = lanes_add<W>(run, v);
        }
    }
    __syncthreads();

    // ---- 4: carry in, rows out
    if (task) {
        const v4 cy = csum[k];
        uint8_t* const o = (uint8_t*)a.out + ((uint64_t)chunk * a.chunk_len + (uint64_t)b * blk) * ESZ + p * 16u;
        // (the carry is the same for the 8 rows: its two halves of the carry-isolated byte add are split once)
        uint32_t cl[4], ch[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { cl[j] = cy[j] & 0x7f7f7f7fu; ch[j] = cy[j] & 0x80808080u; }
#pragma unroll
        for (int r = 0; r < 8; r++) {
            v4 v;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if constexpr (W == 8) v[j] = ((acc[r][j] & 0x7f7f7f7fu) + cl[j]) ^ ((acc[r][j] & 0x80808080u) ^ ch[j]);
                else v[j] = lanes_add<W>(acc[r][j], cy[j]);
            }
            __builtin_nontemporal_store(v, (v4*)(o + (size_t)r * rowbytes));
        }
    }
    // the verbatim tail (:1171): bytes of the image at any phase -> the 16-byte aligned end of the decoded blocks
    if (exists && !corrupt) {
        const uint32_t tb = remaining * ESZ;
        uint8_t* const o = (uint8_t*)a.out + ((uint64_t)chunk * a.chunk_len + (uint64_t)nbo * blk) * ESZ;
        const uint32_t ta = img_a + phase + tail_pos;
        for (uint32_t i = k; i < tb >> 4; i += g.T) {
            v4 v;
            v.x = lds_rd32(ta + 16u * i);
            v.y = lds_rd32(ta + 16u * i + 4u);
            v.z = lds_rd32(ta + 16u * i + 8u);
            v.w = lds_rd32(ta + 16u * i + 12u);
            *(v4*)(o + 16u * i) = v;
        }
        for (uint32_t i = (tb & ~15u) + k; i < tb; i += g.T) o[i] = (uint8_t)lds_rd8(ta + i);
    }
    if (exists && k == 0 && a.rets) a.rets[chunk] = corrupt ? kErrCorrupt : (int64_t)nbo * blk + remaining;
}

hipError_t launch_decode_blk(int w, unsigned grid, hipStream_t st, const DecodeArgs& a, const BlkDecGeom& g);

}  // namespace sprintz

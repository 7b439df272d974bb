// encode_blk.h -- the BLOCK-PARALLEL encoder of the DELTA codec, general (row-major) payload layout, for large batches.
// Same stream bytes, sizes and return values as encode_kernel.h / encode_wide.h / encode_lat.h
// (compress_rowmajor_delta_rle, sprintz_delta_rle.cpp:55-404).
//
// Why a kernel of its own ("next" 1): the lane-per-column encoders walk a chunk's rows in order because the FIRE
// forecast is a recurrence down the column.  Plain delta has none -- a sample's error is x[r][c] - x[r-1][c], element-wise --
// so the only things a block needs from the blocks before it are WHERE its bytes go (a prefix sum of block sizes) and the
// run-length state (a function of which blocks are all zero): two short scans.  So:
//   * a TASK is (chunk, block of 8 rows, 16-byte piece of the row) = 16 columns of uint8 / 8 of uint16: one thread loads its nine
//     16-byte row pieces (the row before the block included), subtracts, zigzags and ORs them four (two) samples to a dword
//     (carry-isolated byte arithmetic / packed 16-bit instructions): ~3 instructions a sample instead of ~12;
//   * the widths of a task's columns stay in registers; the pieces of a block meet through a 2-byte word each in LDS (row bits,
//     the piece's bit offset inside the row);
//   * the RLE / group state machine runs as wave scans over the blocks' "all zero?" bits and sizes (encode_lat.h's formulation,
//     here on 16 / 32 / 64 lanes a chunk, several chunks a wavefront);
//   * fields are packed four (two) to a dword in registers and OR-ed into a zeroed LDS image of the chunk's stream
//     (ds_or_b32, two dwords per 32-bit piece), the verbatim tail is OR-ed in behind, the image leaves in 16-byte stores.
// A workgroup of 256 threads takes as many whole chunks as it has threads for (tasks per chunk <= 256: chunks of at most ~25 KB
// at 80 columns), so a block's samples stay in its thread's registers from the load to the last OR.
//
// Shapes it takes (api.hip: encode_blk_fits): delta codec, general layout, row bytes a multiple of 16, chunk bytes a multiple of
// 16, 16-byte aligned source, <= 256 tasks a chunk, the chunk images of a workgroup within 64 KB of LDS.
#pragma once

#include "decode_fast.h"
#include "encode_kernel.h"
#include "group_ops.h"

namespace sprintz {

struct BlkEncGeom {
    uint32_t P;          // 16-byte pieces per row
    uint32_t NBC;        // whole blocks of a full chunk
    uint32_t T;          // tasks per chunk = NBC * P (<= 256)
    uint32_t CPW;        // chunks per workgroup
    uint32_t GW;         // lanes per chunk in the walk: 16 / 32 / 64
    uint32_t img_cap;    // bytes of one chunk's stream image (multiple of 16; >= compress_bound + 16)
    uint32_t o_psum, o_rbits, o_wofs, o_info, total;     // LDS carve (bytes)
    uint32_t ok;
};

inline BlkEncGeom blk_enc_geom(uint32_t esz, uint32_t chunk_len, uint32_t D, uint32_t bound_bytes)
{
    BlkEncGeom g{};
    const uint32_t rowbytes = D * esz;
    if (rowbytes % 16u || ((uint64_t)chunk_len * esz) % 16u || chunk_len < 16u * D) return g;
    g.P = rowbytes / 16u;
    g.NBC = chunk_len / (8u * D);
    g.T = g.NBC * g.P;
    if (g.T == 0 || g.T > 256u || g.NBC >= 32767u) return g;
    g.GW = g.NBC > 32u ? 64u : g.NBC > 16u ? 32u : 16u;
    const uint32_t by_tasks = 256u / g.T, by_walk = 4u * (64u / g.GW);
    g.img_cap = (bound_bytes + 16u + 15u) & ~15u;
    auto al = [](uint32_t x) { return (x + 15u) & ~15u; };
    uint32_t cpw = by_tasks < by_walk ? by_tasks : by_walk;
    for (; cpw >= 1; cpw--) {
        g.CPW = cpw;
        g.o_psum = cpw * g.img_cap;
        g.o_rbits = g.o_psum + al(cpw * g.T * 2u);
        g.o_wofs = g.o_rbits + al(cpw * g.NBC * 4u);
        g.o_info = g.o_wofs + al(cpw * g.NBC * 8u);
        g.total = g.o_info + cpw * 16u;
        if (g.total <= 64u * 1024u) break;
    }
    g.ok = cpw >= 1 ? 1u : 0u;
    return g;
}

// ---- sample arithmetic, four 8-bit / two 16-bit samples to a dword
// zigzag(x - y) per element (sprintz_delta_rle.cpp:197-205, bitpack.h:302-303)
template <int W> __device__ __forceinline__ uint32_t zz_delta(uint32_t x, uint32_t y)
{
    if constexpr (W == 8) {
        constexpr uint32_t H = 0x80808080u;
        // bytewise x - y: the top bit of every byte is kept out of the subtraction (it cannot borrow from its neighbour) and put back by XOR
        const uint32_t d = ((x | H) - (y & ~H)) ^ ((x ^ ~y) & H);
        // bytewise zigzag: (d << 1) ^ (d >> 7, arithmetic).  The sign masks come from v_perm_b32's sign selectors (8 .. 11 replicate bit 15 / 31
        // of a source): with d << 8 as the second source every byte's sign bit sits at one of those four positions
        const uint32_t sm = __builtin_amdgcn_perm(d, d << 8, 0x0b090a08u);
        return ((d << 1) & 0xfefefefeu) ^ sm;
    } else {
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
        typedef short ss2 __attribute__((ext_vector_type(2)));
        const us2 d = __builtin_bit_cast(us2, x) - __builtin_bit_cast(us2, y);
        const ss2 sg = __builtin_bit_cast(ss2, d) >> 15;
        const us2 z = (d << 1) ^ __builtin_bit_cast(us2, sg);
        return __builtin_bit_cast(uint32_t, z);
    }
}

// width of field f of a dword of OR-ed zigzag values (general layout: 7 -> 8, and 15 -> 16 at 16 bits; sprintz_delta_rle.cpp:259-265)
template <int W> __device__ __forceinline__ uint32_t width_of(uint32_t m, int f)
{
    if constexpr (W == 8) {
        // (the 7 -> 8 rounding was folded into the mask: bit 6 set => bit 7 set)  bit length of a byte = exponent of its float
        const float v = (float)((m >> (8 * f)) & 0xffu);          // v_cvt_f32_ubyte<f>
        return (uint32_t)__builtin_amdgcn_frexp_expf(v);           // 0 for 0
    } else {
        const uint32_t h = f ? m >> 16 : m & 0xffffu;
        const uint32_t nb = 32u - (uint32_t)__clz((int)h);
        return (nb == 7u || nb == 15u) ? nb + 1u : nb;
    }
}

// the low `nb` (<= 64 after the shift: v < 2^32 for payload pieces, < 2^48 for header fields) bits of v OR-ed into the image at bit position bp
__device__ __forceinline__ void img_or32(uint32_t img_a, uint32_t bp, uint32_t v)
{
    typedef __attribute__((address_space(3))) uint32_t lds_word;
    const uint64_t x = (uint64_t)v << (bp & 31u);
    lds_word* q = (lds_word*)(uintptr_t)(img_a + ((bp >> 3) & ~3u));
    __hip_atomic_fetch_or(q, (uint32_t)x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_or(q + 1, (uint32_t)(x >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void img_or64(uint32_t img_a, uint32_t bp, uint64_t v)
{
    typedef __attribute__((address_space(3))) uint32_t lds_word;
    const uint32_t s = bp & 31u;
    const uint64_t lo = v << s;
    const uint32_t top = (uint32_t)((v >> 1) >> (63u - s));
    lds_word* q = (lds_word*)(uintptr_t)(img_a + ((bp >> 3) & ~3u));
    __hip_atomic_fetch_or(q, (uint32_t)lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_or(q + 1, (uint32_t)(lo >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_or(q + 2, top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ---- the RLE / group state machine as scans over GW lanes (encode_lat.h, phase 4 -- the same decisions, see the comment there):
// rbits[b] = row bits of block b (0: all zero).  Writes wofs[b] = (byte offset of the block's payload in the stream, bit position of its
// header fields) for every packed block (blocks inside runs keep 0xffffffff), the run lengths into the image, and
// info = {groups, elements consumed, bytes written before the tail}.  Every lane of the wavefront calls it (shuffles inside).
// LOW: low-dim payload (row bits = payload bytes, no row padding); TAIL_LE: "<=" in the run test (general FIRE codec only).
template <int GW, bool LOW, bool TAIL_LE>
__device__ __forceinline__ void rle_walk_scan(uint32_t lg, bool live, uint32_t n, uint32_t blk, uint32_t NB_in, uint32_t hdr_bytes, uint32_t slot_bits,
                                              uint32_t base_wl, const uint32_t* rbits, uint2* wofs, uint8_t* img, uint32_t* info)
{
    const int64_t limit = (int64_t)n - 2 * (int64_t)blk;        // last_full_group_start (:158), in elements
    const bool coded = live && n >= 128u && limit >= 0;            // :116 and the loop guard :160: otherwise header + verbatim samples
    const uint32_t NB = coded ? NB_in : 0u;
    const int32_t lim_le = limit >= 0 ? (int32_t)(limit / (int64_t)blk) : -1;              // pos_in <= limit  <=>  b <= lim_le
    const int32_t lim_lt = limit > 0 ? (int32_t)((limit - 1) / (int64_t)blk) : -1;         // pos_in <  limit  <=>  b <= lim_lt
    const int32_t lim_run = TAIL_LE ? lim_le : lim_lt;
    const uint32_t P = (NB + (uint32_t)GW - 1u) / (uint32_t)GW;
    const uint32_t i0 = lg * P < NB ? lg * P : NB, i1 = i0 + P < NB ? i0 + P : NB;
    auto payload = [&](uint32_t rb) -> uint32_t { return LOW ? rb : ((rb + 7u) >> 3) << 3; };
    auto rb_at = [&](uint32_t bb) -> uint32_t { return bb < NB ? rbits[bb] : 1u; };
    const bool first_prev = (i0 == 0 || i0 >= NB) ? true : rbits[i0 - 1] != 0;      // (block 0 starts a slot whatever it is)
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
    const uint32_t starts_before = group_scan<GW>(starts, (int)lg, starts_total);
    int lnz_before;                                          // exclusive max-scan: the last packed block before the piece
    {
        int incl = lnz;
        for (int off = 1; off < GW; off <<= 1) { const int t = __shfl_up(incl, off, GW); if ((int)lg >= off) incl = incl > t ? incl : t; }
        lnz_before = __shfl_up(incl, 1, GW);
        if (lg == 0) lnz_before = -1;
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
    for (int off = GW >> 1; off > 0; off >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)cand, off, GW); cand = t < cand ? t : cand; }
    const uint32_t stop_b = cand < NB ? cand : NB - 1u;      // (it always exists: the last two blocks trip one of the tests; NB == 0: no block at all)
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
            else if (bb == stop_b || rb_at(bb + 1u) != 0) bytes += ((int)bb - last) > 127 ? 2u : 1u;
            prev = nz;
        }
    }
    uint32_t bytes_total;
    const uint32_t bytes_before = group_scan<GW>(bytes, (int)lg, bytes_total);
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
            else if (bb == stop_b || rb_at(bb + 1u) != 0) off += ((int)bb - last) > 127 ? 2u : 1u;
            prev = nz;
        }
    }
    uint32_t hp_before;
    {
        uint32_t incl = hp;
        for (int off = 1; off < GW; off <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, off, GW); if ((int)lg >= off) incl = incl > t ? incl : t; }
        hp_before = (uint32_t)__shfl_up((int)incl, 1, GW);
        if (lg == 0) hp_before = 0;
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
            } else if (bb == stop_b || rb_at(bb + 1u) != 0) {
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
    if (live && !coded && lg == 0) { info[0] = 0; info[1] = 0; info[2] = base_wl; }
}

template <int W>
__global__ void __launch_bounds__(256) encode_blk_kernel(EncodeArgs a, BlkEncGeom g)
{
    constexpr int HB = Elem<W>::HB;
    constexpr int ESZ = W / 8;
    constexpr int CPP = 16 / ESZ;                // columns per piece
    constexpr int FPD = 4 / ESZ;                 // fields per dword
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const uint32_t tid = threadIdx.x;
    const uint32_t wgn = blockIdx.x;
    const uint32_t ci = tid / g.T, k = tid - ci * g.T;
    const uint32_t b = k / g.P, p = k - b * g.P;
    const bool in_wg = ci < g.CPW;
    const uint64_t chunk = (uint64_t)wgn * g.CPW + ci;
    const bool exists = in_wg && chunk < a.nchunks;
    const uint32_t D = (uint32_t)a.D, blk = 8u * D, rowbytes = D * ESZ;
    const uint64_t first = chunk * (uint64_t)a.chunk_len;
    const uint32_t n = exists ? (uint32_t)((a.total_len - first < a.chunk_len) ? (a.total_len - first) : a.chunk_len) : 0u;
    const uint32_t NB = n / blk;                 // whole blocks this chunk holds
    const bool task = exists && b < NB;
    const uint32_t cix = in_wg ? ci : 0u;        // (threads past the last chunk of the workgroup: addresses stay inside the carve)
    uint8_t* const img = smem + cix * g.img_cap;
    const uint32_t img_a = lds_addr(img);
    uint16_t* const psum = (uint16_t*)(smem + g.o_psum) + cix * g.T;
    uint32_t* const rbits = (uint32_t*)(smem + g.o_rbits) + cix * g.NBC;
    uint2* const wofs = (uint2*)(smem + g.o_wofs) + cix * g.NBC;
    uint32_t* const info = (uint32_t*)(smem + g.o_info) + cix * 4u;
    const uint8_t* const csrc = (const uint8_t*)a.src + first * ESZ;

    // ---- A: the block's nine row pieces (the row in front of block 0 is zero: state resets per chunk, :61-63)
    v4 x[9];
#pragma unroll
    for (int r = 0; r < 9; r++) x[r] = v4{0u, 0u, 0u, 0u};
    if (task) {
        const uint8_t* const s = csrc + (size_t)b * blk * ESZ + p * 16u;
        if (b != 0) x[0] = *(const v4*)(s - rowbytes);
#pragma unroll
        for (int r = 0; r < 8; r++) x[r + 1] = *(const v4*)(s + (size_t)r * rowbytes);
    }
    // (while the loads are in flight) clean images
    for (uint32_t i = tid; i < (g.CPW * g.img_cap) >> 4; i += 256u) ((uint4*)smem)[i] = make_uint4(0, 0, 0, 0);

    uint32_t z[8][4], m[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int r = 0; r < 8; r++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            z[r][j] = zz_delta<W>(x[r + 1][j], x[r][j]);
            m[j] |= z[r][j];
        }
    uint32_t nb[4][FPD], wsum[4], S = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        uint32_t mj = m[j];
        if constexpr (W == 8) mj |= (mj & 0x40404040u) << 1;       // 7 -> 8: a byte with bit 6 set gets bit 7
        wsum[j] = 0;
#pragma unroll
        for (int f = 0; f < FPD; f++) { nb[j][f] = width_of<W>(mj, f); wsum[j] += nb[j][f]; }
        S += wsum[j];
    }
    if (task) psum[k] = (uint16_t)S;
    __syncthreads();

    // ---- the pieces of a block meet: row bits, this piece's bit offset inside the row
    uint32_t Bp = 0, tot = 0;
    if (task) {
        for (uint32_t q = 0; q < g.P; q++) {
            const uint32_t s = psum[b * g.P + q];
            Bp += q < p ? s : 0u;
            tot += s;
        }
        if (p == 0) { rbits[b] = tot; wofs[b] = make_uint2(0xffffffffu, 0u); }
    }
    __syncthreads();

    // ---- B: the RLE / group state machine: wavefront w takes the chunks [w * 64 / GW, (w + 1) * 64 / GW) of the workgroup, GW lanes each
    {
        const uint32_t wave = tid >> 6, lane = tid & 63u;
        const uint32_t cpwave = 64u / g.GW;
        if (wave * cpwave < g.CPW) {                                     // (uniform per wavefront)
            const uint32_t wc = wave * cpwave + lane / g.GW, lg = lane & (g.GW - 1u);
            const bool wlive = wc < g.CPW && (uint64_t)wgn * g.CPW + wc < a.nchunks;
            const uint32_t wcx = wc < g.CPW ? wc : 0u;
            const uint64_t wfirst = ((uint64_t)wgn * g.CPW + wc) * (uint64_t)a.chunk_len;
            const uint32_t wn = wlive ? (uint32_t)((a.total_len - wfirst < a.chunk_len) ? (a.total_len - wfirst) : a.chunk_len) : 0u;
            const uint32_t hdr_bytes = (2u * D * HB + 7u) >> 3;
            const uint32_t* const wr = (const uint32_t*)(smem + g.o_rbits) + wcx * g.NBC;
            uint2* const ww = (uint2*)(smem + g.o_wofs) + wcx * g.NBC;
            uint8_t* const wi = smem + wcx * g.img_cap;
            uint32_t* const wf = (uint32_t*)(smem + g.o_info) + wcx * 4u;
            const uint32_t base_wl = a.write_size ? 8u : 0u;
            if (g.GW == 16u) rle_walk_scan<16, false, false>(lg, wlive, wn, blk, wn / blk, hdr_bytes, D * HB, base_wl, wr, ww, wi, wf);
            else if (g.GW == 32u) rle_walk_scan<32, false, false>(lg, wlive, wn, blk, wn / blk, hdr_bytes, D * HB, base_wl, wr, ww, wi, wf);
            else rle_walk_scan<64, false, false>(lg, wlive, wn, blk, wn / blk, hdr_bytes, D * HB, base_wl, wr, ww, wi, wf);
        }
    }
    __syncthreads();
    const uint32_t ngroups = info[0], pos_in = info[1], wl = info[2];
    const uint32_t remaining = n - pos_in;

    // ---- C: header fields (:296) and payload rows (:483-524) into the image
    if (task) {
        const uint2 wo = wofs[b];
        if (wo.x != 0xffffffffu) {
            uint64_t hv = 0;
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int f = 0; f < FPD; f++) {
                    const uint32_t code = nb[j][f] == (uint32_t)W ? (uint32_t)(W - 1) : nb[j][f];
                    hv |= (uint64_t)code << (HB * (j * FPD + f));
                }
            img_or64(img_a, wo.y + p * (uint32_t)(CPP * HB), hv);
            const uint32_t row_bits = ((tot + 7u) >> 3) << 3;
            uint32_t sh[4][FPD];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                sh[j][0] = 0;
#pragma unroll
                for (int f = 1; f < FPD; f++) sh[j][f] = sh[j][f - 1] + nb[j][f - 1];
            }
            uint32_t rowbit = wo.x * 8u + Bp;
#pragma unroll
            for (int r = 0; r < 8; r++) {
                uint32_t bp = rowbit;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    uint32_t t;
                    if constexpr (W == 8)
                        t = (z[r][j] & 0xffu) | (((z[r][j] >> 8) & 0xffu) << sh[j][1]) | (((z[r][j] >> 16) & 0xffu) << sh[j][2]) | ((z[r][j] >> 24) << sh[j][3]);
                    else
                        t = (z[r][j] & 0xffffu) | ((z[r][j] >> 16) << sh[j][1]);
                    img_or32(img_a, bp, t);
                    bp += wsum[j];
                }
                rowbit += row_bits;
            }
        }
    }
    // ---- the verbatim tail (:553) behind the coded part: dwords of the source OR-ed in at the image's byte phase
    if (exists) {
        const uint32_t tb = remaining * ESZ;                                 // bytes
        const uint8_t* const tp = csrc + (size_t)pos_in * ESZ;               // 4-byte aligned: pos_in is whole blocks, rows are multiples of 16 bytes
        for (uint32_t i = k; i < (tb + 3u) >> 2; i += g.T) {
            uint32_t v = ((const uint32_t*)tp)[i];                           // (the last dword may reach past the chunk: inside SPRINTZ_MI355X_READ_SLACK)
            const uint32_t left = tb - 4u * i;
            if (left < 4u) v &= (1u << (8u * left)) - 1u;
            img_or32(img_a, (wl + 4u * i) * 8u, v);
        }
    }
    __syncthreads();
    if (exists && k == 0 && a.write_size) {                                  // (bytes 0 .. 7 are nobody else's: plain stores)
        ((uint32_t*)img)[0] = ngroups;
        ((uint32_t*)img)[1] = (remaining & 0xffffu) | (D << 16);
    }
    __syncthreads();

    // ---- the stream leaves in 16-byte pieces (the slot is 16-byte aligned and holds the bound)
    if (exists) {
        const uint32_t total_bytes = wl + remaining * ESZ;
        uint8_t* const gdst = a.slots + chunk * a.slot_stride;
        for (uint32_t i = k; i < (total_bytes + 15u) >> 4; i += g.T) ((uint4*)gdst)[i] = ((const uint4*)img)[i];
        if (k == 0) {
            a.sizes[chunk] = total_bytes;
            if (a.rets) a.rets[chunk] = (int64_t)(total_bytes / ESZ);
        }
    }
}

// ---- the same scheme for UNIVARIATE streams of the low-dim layout (compress_rowmajor_delta_rle_lowdim, ndims == 1:
// sprintz_delta_lowdim.cpp:39-384; BASELINE config 1): a task is 16 bytes of the series = two blocks of uint8 / one of uint16; a
// block's payload is its 8 fields back to back = nbits BYTES, its header field the 3 / 4 bits of its slot; widths are
// rounded W - 1 -> W only (:207-208).  g: P = 1, T = tasks a chunk, NBC = blocks a chunk (blk_enc_uni_geom).
inline BlkEncGeom blk_enc_uni_geom(uint32_t esz, uint32_t chunk_len, uint32_t bound_bytes)
{
    BlkEncGeom g{};
    if (((uint64_t)chunk_len * esz) % 16u || chunk_len < 16u) return g;
    g.P = 1;
    g.NBC = chunk_len / 8u;
    g.T = chunk_len * esz / 16u;
    if (g.T == 0 || g.T > 256u) return g;
    // (BASELINE config 1, 128 blocks a chunk: 0.565 ms with 64 lanes a chunk in the walk, 0.661 with 16 -- four chunks a walking wavefront, eight
    //  blocks a lane: the walk's price is its passes over a lane's blocks, not its wave-wide scans; the lane-per-chunk kernel takes 0.399)
    g.GW = g.NBC > 32u ? 64u : g.NBC > 16u ? 32u : 16u;
    const uint32_t by_tasks = 256u / g.T, by_walk = 4u * (64u / g.GW);
    g.img_cap = (bound_bytes + 16u + 15u) & ~15u;
    auto al = [](uint32_t x) { return (x + 15u) & ~15u; };
    uint32_t cpw = by_tasks < by_walk ? by_tasks : by_walk;
    for (; cpw >= 1; cpw--) {
        g.CPW = cpw;
        g.o_psum = cpw * g.img_cap;                      // (unused: one column)
        g.o_rbits = g.o_psum;
        g.o_wofs = g.o_rbits + al(cpw * g.NBC * 4u);
        g.o_info = g.o_wofs + al(cpw * g.NBC * 8u);
        g.total = g.o_info + cpw * 16u;
        if (g.total <= 64u * 1024u) break;
    }
    g.ok = cpw >= 1 ? 1u : 0u;
    return g;
}

template <int W>
__global__ void __launch_bounds__(256) encode_blk_uni_kernel(EncodeArgs a, BlkEncGeom g)
{
    constexpr int HB = Elem<W>::HB;
    constexpr int ESZ = W / 8;
    constexpr int BPT = 2 / ESZ;                 // blocks per task: 16 bytes = 2 x 8 uint8 / 1 x 8 uint16
    constexpr int DPB = 4 / BPT;                 // dwords per block
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const uint32_t tid = threadIdx.x;
    const uint32_t ci = tid / g.T, k = tid - ci * g.T;
    const bool in_wg = ci < g.CPW;
    const uint64_t chunk = (uint64_t)blockIdx.x * g.CPW + ci;
    const bool exists = in_wg && chunk < a.nchunks;
    const uint64_t first = chunk * (uint64_t)a.chunk_len;
    const uint32_t n = exists ? (uint32_t)((a.total_len - first < a.chunk_len) ? (a.total_len - first) : a.chunk_len) : 0u;
    const uint32_t NB = n / 8u;                  // whole blocks this chunk holds
    const uint32_t b0 = k * BPT;                 // the task's first block
    const bool task = exists && b0 < NB;         // (its second block may lie past the last whole one: checked per block)
    const uint32_t cix = in_wg ? ci : 0u;
    uint8_t* const img = smem + cix * g.img_cap;
    const uint32_t img_a = lds_addr(img);
    uint32_t* const rbits = (uint32_t*)(smem + g.o_rbits) + cix * g.NBC;
    uint2* const wofs = (uint2*)(smem + g.o_wofs) + cix * g.NBC;
    uint32_t* const info = (uint32_t*)(smem + g.o_info) + cix * 4u;
    const uint8_t* const csrc = (const uint8_t*)a.src + first * ESZ;

    // ---- A: 16 bytes of the series and the sample in front of them (0 at the chunk's start: state resets per chunk)
    v4 x = v4{0u, 0u, 0u, 0u};
    uint32_t prev = 0;
    if (task) {
        x = *(const v4*)(csrc + 16u * k);                                // (the chunk's last piece may reach past it: inside SPRINTZ_MI355X_READ_SLACK)
        if (k != 0) prev = ESZ == 1 ? (uint32_t)csrc[16u * k - 1u] << 24 : (uint32_t)((const uint16_t*)csrc)[8u * k - 1u] << 16;
    }
    for (uint32_t i = tid; i < (g.CPW * g.img_cap) >> 4; i += 256u) ((uint4*)smem)[i] = make_uint4(0, 0, 0, 0);

    uint32_t z[4];
    {
        uint32_t before = prev;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t xs = __builtin_amdgcn_alignbyte(x[j], before, 4 - ESZ);       // the series shifted by one sample
            z[j] = zz_delta<W>(x[j], xs);
            before = x[j];
        }
    }
    uint32_t nbk[BPT];
#pragma unroll
    for (int q = 0; q < BPT; q++) {
        uint32_t m = 0;
#pragma unroll
        for (int d = 0; d < DPB; d++) m |= z[q * DPB + d];
        m |= m >> 16;
        if constexpr (W == 8) m |= m >> 8;
        m &= Elem<W>::MASK;
        nbk[q] = nbits_of<W, true>(m);                                   // W - 1 -> W only
        if (task && b0 + (uint32_t)q < NB) { rbits[b0 + q] = nbk[q]; wofs[b0 + q] = make_uint2(0xffffffffu, 0u); }
    }
    __syncthreads();

    // ---- B: the RLE / group state machine (payload bytes of a block = its width: one column)
    {
        const uint32_t wave = tid >> 6, lane = tid & 63u;
        const uint32_t cpwave = 64u / g.GW;
        if (wave * cpwave < g.CPW) {
            const uint32_t wc = wave * cpwave + lane / g.GW, lg = lane & (g.GW - 1u);
            const bool wlive = wc < g.CPW && (uint64_t)blockIdx.x * g.CPW + wc < a.nchunks;
            const uint32_t wcx = wc < g.CPW ? wc : 0u;
            const uint64_t wfirst = ((uint64_t)blockIdx.x * g.CPW + wc) * (uint64_t)a.chunk_len;
            const uint32_t wn = wlive ? (uint32_t)((a.total_len - wfirst < a.chunk_len) ? (a.total_len - wfirst) : a.chunk_len) : 0u;
            const uint32_t hdr_bytes = (2u * HB + 7u) >> 3;
            const uint32_t* const wr = (const uint32_t*)(smem + g.o_rbits) + wcx * g.NBC;
            uint2* const ww = (uint2*)(smem + g.o_wofs) + wcx * g.NBC;
            uint8_t* const wi = smem + wcx * g.img_cap;
            uint32_t* const wf = (uint32_t*)(smem + g.o_info) + wcx * 4u;
            const uint32_t base_wl = a.write_size ? 8u : 0u;
            if (g.GW == 16u) rle_walk_scan<16, true, false>(lg, wlive, wn, 8u, wn / 8u, hdr_bytes, HB, base_wl, wr, ww, wi, wf);
            else if (g.GW == 32u) rle_walk_scan<32, true, false>(lg, wlive, wn, 8u, wn / 8u, hdr_bytes, HB, base_wl, wr, ww, wi, wf);
            else rle_walk_scan<64, true, false>(lg, wlive, wn, 8u, wn / 8u, hdr_bytes, HB, base_wl, wr, ww, wi, wf);
        }
    }
    __syncthreads();
    const uint32_t ngroups = info[0], pos_in = info[1], wl = info[2];
    const uint32_t remaining = n - pos_in;

    // ---- C: a block's header field and its 8 fields, nbits bytes, into the image (sprintz_delta_lowdim.cpp:306-357)
    if (task) {
#pragma unroll
        for (int q = 0; q < BPT; q++) {
            if (b0 + (uint32_t)q >= NB) continue;
            const uint2 wo = wofs[b0 + q];
            if (wo.x == 0xffffffffu) continue;
            const uint32_t nb = nbk[q];
            img_or32(img_a, wo.y, nb == (uint32_t)W ? (uint32_t)(W - 1) : nb);
            if constexpr (W == 8) {
                const uint32_t za = z[2 * q], zb = z[2 * q + 1];
                const uint32_t lo = (za & 0xffu) | (((za >> 8) & 0xffu) << nb) | (((za >> 16) & 0xffu) << (2u * nb)) | ((za >> 24) << (3u * nb));
                const uint32_t hi = (zb & 0xffu) | (((zb >> 8) & 0xffu) << nb) | (((zb >> 16) & 0xffu) << (2u * nb)) | ((zb >> 24) << (3u * nb));
                img_or64(img_a, wo.x * 8u, (uint64_t)lo | ((uint64_t)hi << (4u * nb)));           // 4 nb <= 32
            } else {
                // fields 0 .. 3 (<= 64 bits) and 4 .. 7, 4 nb bits further on
                uint64_t v[2];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const uint32_t za = z[2 * h], zb = z[2 * h + 1];
                    const uint64_t p0 = (uint64_t)(za & 0xffffu) | ((uint64_t)(za >> 16) << nb);   // <= 32 bits
                    const uint64_t p1 = (uint64_t)(zb & 0xffffu) | ((uint64_t)(zb >> 16) << nb);
                    v[h] = p0 | (p1 << (2u * nb));                                               // 2 nb <= 32
                }
                img_or64(img_a, wo.x * 8u, v[0]);
                img_or64(img_a, wo.x * 8u + 4u * nb, v[1]);
            }
        }
    }
    if (exists) {                                                            // the verbatim tail (:553), dwords OR-ed in at the image's byte phase
        const uint32_t tb = remaining * ESZ;
        const uint8_t* const tp = csrc + (size_t)pos_in * ESZ;               // pos_in is whole blocks: 8 / 16-byte aligned
        for (uint32_t i = k; i < (tb + 3u) >> 2; i += g.T) {
            uint32_t v = ((const uint32_t*)tp)[i];
            const uint32_t left = tb - 4u * i;
            if (left < 4u) v &= (1u << (8u * left)) - 1u;
            img_or32(img_a, (wl + 4u * i) * 8u, v);
        }
    }
    __syncthreads();
    if (exists && k == 0 && a.write_size) {
        ((uint32_t*)img)[0] = ngroups;
        ((uint32_t*)img)[1] = (remaining & 0xffffu) | (1u << 16);
    }
    __syncthreads();
    if (exists) {
        const uint32_t total_bytes = wl + remaining * ESZ;
        uint8_t* const gdst = a.slots + chunk * a.slot_stride;
        for (uint32_t i = k; i < (total_bytes + 15u) >> 4; i += g.T) ((uint4*)gdst)[i] = ((const uint4*)img)[i];
        if (k == 0) {
            a.sizes[chunk] = total_bytes;
            if (a.rets) a.rets[chunk] = (int64_t)(total_bytes / ESZ);
        }
    }
}

hipError_t launch_encode_blk(int w, unsigned grid, hipStream_t st, const EncodeArgs& a, const BlkEncGeom& g);
hipError_t launch_encode_blk_uni(int w, unsigned grid, hipStream_t st, const EncodeArgs& a, const BlkEncGeom& g);

}  // namespace sprintz

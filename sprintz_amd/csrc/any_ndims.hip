// any_ndims.hip -- the general layout for streams of 513 .. 65 535 columns (2 048 and up: the "big" kernels at the end of the file).  513 .. 2047 columns: ONE WORKGROUP (256 lanes) per chunk, lane t owns the
// columns [t * cpl, (t + 1) * cpl), cpl = ceil(ndims / 256) <= 8.
//
// The reference takes any uint16 ndims that fits the header (format.h:36-45; its tests stop at 129, test/compress_testing.hpp:20-21);
// the lane-group kernels of this library carry at most 64 lanes x 8 columns.  These two kernels are the same codecs
// (sprintz_xff_rle.cpp:61-555 / :569-1179, sprintz_delta_rle.cpp:55-404 / :418-772) written for completeness, not speed: the per-block
// width scan and the slot totals are workgroup-wide (a wave scan + four partial sums through LDS), the encoder's fields are OR-ed
// into an LDS image of ONE stream group (<= 66 KB at 2047 uint16 columns) that is flushed in 16-byte pieces when the next group
// starts, the decoder reads its fields from global memory.  Control flow is uniform across the workgroup (every decision is a
// function of header totals and cursors all lanes hold alike), so the barriers inside the loops are met by all 256 lanes.
#include "launch.h"

namespace sprintz {
namespace {

constexpr int kAnyCpl = 8;

// exclusive prefix of v over the 256 lanes of the workgroup (lane order), and the total; s4: 4 words of LDS
__device__ __forceinline__ uint32_t wg_scan(uint32_t v, uint32_t& total, uint32_t* s4)
{
    const uint32_t tid = threadIdx.x;
    uint32_t wave_total;
    const uint32_t ex = group_scan<64>(v, (int)(tid & 63u), wave_total);
    __syncthreads();                                   // (s4 may still be read from the scan before)
    if ((tid & 63u) == 0) s4[tid >> 6] = wave_total;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4; w++) {
        const uint32_t x = s4[w];
        base += w < (tid >> 6) ? x : 0u;
        tot += x;
    }
    total = tot;
    return ex + base;
}

template <int W, bool FIRE, int CPL>
__global__ void __launch_bounds__(256) decode_any_kernel(DecodeArgs a)
{
    using U = typename Elem<W>::U;
    constexpr int HB = Elem<W>::HB;
    constexpr uint32_t MASK = Elem<W>::MASK;
    constexpr int ESZ = W / 8;
    __shared__ uint32_t s4[4];
    // Round 5: the stream reaches the lanes through LDS and the samples leave through it.  [header image | payload image | block image]:
    // a group's header and a block's 8 rows are copied in with 16-byte requests (round 4 fetched every field's bits from global memory:
    // two dependent dword loads a field), the decoded 8 x D block is assembled row-major in LDS and leaves in 16-byte pieces (round 4:
    // one 2-byte store per sample) -- 1.05 -> 0.91 ms for 1 024 chunks of 1 000 columns: a chunk's 32 blocks are serial phases behind barriers, which is what is left.
    extern __shared__ __attribute__((aligned(16))) uint8_t any_lds[];
    const uint32_t tid = threadIdx.x;
    const uint64_t chunk = blockIdx.x;
    const int D = a.D;
    const int cpl = (D + 255) / 256;
    const int col0 = (int)tid * cpl;
    const uint32_t hdr_cap = ((((2u * (uint32_t)D * HB + 7u) >> 3) + 15u) & ~15u) + 32u;      // + the source's misalignment + the window's over-read
    const uint32_t blk_cap = ((8u * (uint32_t)D * ESZ + 15u) & ~15u) + 32u;
    uint8_t* const l_hdr = any_lds;
    uint8_t* const l_pay = any_lds + hdr_cap;
    uint8_t* const l_out = l_pay + blk_cap;
    // n bytes from global g (any alignment) -> the image at l: returns the image offset of g's first byte (= g's misalignment)
    auto stage_in = [&](const uint8_t* g, uint32_t n, uint8_t* l) -> uint32_t {
        const uint32_t mis = (uint32_t)((uintptr_t)g & 15u);
        const uint8_t* const ga = g - mis;                       // (an aligned 16-byte piece never leaves the lines the bytes themselves lie in)
        for (uint32_t u = tid * 16u; u < mis + n; u += 256u * 16u) *(uint4*)(l + u) = *(const uint4*)(ga + u);
        return mis;
    };
    // n bytes of the image at l (whose byte 0 is the aligned line below g) -> global g: whole 16-byte pieces, the two ends byte by byte
    auto stage_out = [&](uint8_t* g, uint32_t n, const uint8_t* l) {
        const uint32_t mis = (uint32_t)((uintptr_t)g & 15u), end = mis + n;
        uint8_t* const ga = g - mis;
        for (uint32_t u = tid * 16u; u < end; u += 256u * 16u) {
            if (u >= mis && u + 16u <= end) *(uint4*)(ga + u) = *(const uint4*)(l + u);
            else for (uint32_t k = u < mis ? mis : u; k < u + 16u && k < end; k++) ga[k] = l[k];
        }
    };

    const uint64_t off_c = a.offsets[chunk];
    const uint8_t* const s = a.comp + off_c;
    const uint64_t slen64 = a.offsets[chunk + 1] - off_c;
    const uint32_t stream_len = slen64 < 0xffffffffull ? (uint32_t)slen64 : 0xffffffffu;
    U* const o = (U*)a.out + chunk * (uint64_t)a.chunk_len;

    // ---- 8-byte stream header (format.h:48-62), or the caller's numbers (sprintz_xff.h:56-58)
    uint32_t groups_left, remaining, pos;
    bool corrupt = false;
    if (!a.noheader) {
        if (stream_len < 8u) { corrupt = true; groups_left = 0; remaining = 0; pos = 0; }
        else {
            const uint32_t w0 = load_u32_any(s), w1 = load_u32_any(s + 4);
            groups_left = w0;
            remaining = w1 & 0xffffu;
            pos = 8;
            if ((int)(w1 >> 16) != D) corrupt = true;
        }
    } else {
        groups_left = a.nh_ngroups;
        remaining = a.nh_remaining;
        pos = 0;
    }
    const uint32_t hdr_bytes = (2u * (uint32_t)D * HB + 7u) >> 3;
    const uint32_t blk_elems = 8u * (uint32_t)D;
    if (groups_left > a.chunk_len / blk_elems + 2u) corrupt = true;      // a damaged header must not make the loop spin
    if (corrupt) groups_left = 0;

    uint32_t pv[CPL];
    int pd[CPL], ctr[CPL];
    uint32_t nbs0[CPL], nbs1[CPL];                            // the two slots' widths (one array indexed by the slot went to scratch)
#pragma unroll
    for (int k = 0; k < CPL; k++) { pv[k] = 0; pd[k] = 0; ctr[k] = 0; nbs0[k] = 0; nbs1[k] = 0; }
    uint32_t out_elems = 0;

    // one block of errors z (zigzagged; run blocks: zeros) -> samples, stored (:993-1150; runs :828-958)
    auto emit_block = [&](const uint32_t (&z)[8][CPL], bool run_block) {
        U* const og = o + out_elems;
        const uint32_t omis = (uint32_t)((uintptr_t)og & 15u);
        U* const ob = (U*)(l_out + omis);                         // (element-aligned: the output's misalignment is a multiple of the element size)
#pragma unroll
        for (int k = 0; k < CPL; k++) {
            const int col = col0 + k;
            if (k >= cpl || col >= D) continue;
            int coef = FIRE ? fire_coef<W, false>(ctr[k]) : 0;
            if constexpr (FIRE && W == 16) {
                if (a.quirk && run_block) coef = fire_coef_ref_run16(ctr[k], col);
            }
            int grad = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int err = unzigzag(z[i][k]);
                const int pred = FIRE ? fire_predict<W, false>(pd[k], coef) : 0;
                const int delta = sext<W>(err + pred);
                if (FIRE && (i & 1)) grad += sign_times(err, pd[k]);
                pv[k] = (pv[k] + (uint32_t)delta) & MASK;
                pd[k] = delta;
                ob[(uint32_t)i * (uint32_t)D + (uint32_t)col] = (U)pv[k];
            }
            if (FIRE) ctr[k] = wrap_counter<W>(ctr[k] + (sext<W>(grad) >> 2));       // :1120-1128
        }
        __syncthreads();
        stage_out((uint8_t*)og, blk_elems * ESZ, l_out);
        __syncthreads();
        out_elems += blk_elems;
    };

    while (groups_left > 0 && !corrupt) {
        groups_left--;
        if (hdr_bytes > stream_len - pos) { corrupt = true; break; }
        // ---- group header: 2 D fields of HB bits, LSB first (:713-735)
        const uint8_t* const hsrc = l_hdr + stage_in(s + pos, hdr_bytes, l_hdr);
        __syncthreads();
        uint32_t both = 0;
#pragma unroll
        for (int k = 0; k < CPL; k++) {
            const int col = col0 + k;
            uint32_t f0 = 0, f1 = 0;
            if (k < cpl && col < D) {
                f0 = fetch_bits(hsrc, (uint32_t)col * HB, HB);
                f1 = fetch_bits(hsrc, (uint32_t)(D + col) * HB, HB);
            }
            nbs0[k] = f0 == (uint32_t)(W - 1) ? (uint32_t)W : f0;        // :747-749, :763-765
            nbs1[k] = f1 == (uint32_t)(W - 1) ? (uint32_t)W : f1;
            both += nbs0[k] | (nbs1[k] << 16);                        // (a slot's total is at most 2047 * 16 < 2^16)
        }
        uint32_t tot_both;
        const uint32_t excl_both = wg_scan(both, tot_both, s4);
        pos += hdr_bytes;
        for (int slot = 0; slot < 2 && !corrupt; slot++) {
            const uint32_t total = slot ? tot_both >> 16 : tot_both & 0xffffu;
            if (total == 0) {                                             // RUN slot: varint length in blocks (:829-833)
                if (stream_len - pos < 2u) {
                    if (stream_len == pos || (load_u8(s + pos) & 0x80u)) { corrupt = true; break; }
                }
                const uint32_t b0 = load_u8(s + pos);
                uint32_t len = b0 & 0x7fu;
                if (b0 & 0x80u) { len |= load_u8(s + pos + 1) << 7; pos += 2; }
                else pos += 1;
                uint32_t zero[8][CPL];
#pragma unroll
                for (int i = 0; i < 8; i++)
#pragma unroll
                    for (int k = 0; k < CPL; k++) zero[i][k] = 0;
                for (; len > 0; len--) {
                    if (out_elems + blk_elems > a.chunk_len) { corrupt = true; break; }
                    emit_block(zero, true);
                }
            } else {                                                      // packed block: 8 rows of ceil(total / 8) bytes (:961-990)
                const uint32_t row_bits = ((total + 7u) >> 3) << 3;
                if (row_bits > stream_len - pos || out_elems + blk_elems > a.chunk_len) { corrupt = true; break; }
                uint32_t off = slot ? excl_both >> 16 : excl_both & 0xffffu;
                const uint8_t* const psrc = l_pay + stage_in(s + pos, row_bits, l_pay);      // (row_bits = the block's 8 rows in BYTES)
                __syncthreads();
                uint32_t z[8][CPL];
#pragma unroll
                for (int k = 0; k < CPL; k++) {
                    const uint32_t nb = slot ? nbs1[k] : nbs0[k];
#pragma unroll
                    for (int i = 0; i < 8; i++)
                        z[i][k] = (k < cpl && col0 + k < D) ? fetch_bits(psrc, (uint32_t)i * row_bits + off, nb) : 0u;
                    off += nb;
                }
                emit_block(z, false);
                pos += row_bits;
            }
        }
    }

    // ---- verbatim tail (:1171)
    if (!corrupt && (out_elems + remaining > a.chunk_len || (uint64_t)remaining * ESZ > (uint64_t)(stream_len - pos))) corrupt = true;
    if (!corrupt) {
        const uint8_t* const t = s + pos;
        uint8_t* const d = (uint8_t*)(o + out_elems);
        for (uint32_t j = tid; j < remaining * ESZ; j += 256u) d[j] = t[j];
    }
    if (tid == 0 && a.rets) a.rets[chunk] = corrupt ? kErrCorrupt : (int64_t)out_elems + remaining;
}

template <int W, bool FIRE, int CPL>
__global__ void __launch_bounds__(256) encode_any_kernel(EncodeArgs a)
{
    using U = typename Elem<W>::U;
    typedef __attribute__((address_space(3))) uint32_t lds_word;
    constexpr int HB = Elem<W>::HB;
    constexpr int ESZ = W / 8;
    constexpr bool TAIL_LE = FIRE;               // "<=" at sprintz_xff_rle.cpp:362, "<" at sprintz_delta_rle.cpp:226
    extern __shared__ __attribute__((aligned(16))) uint8_t win[];        // the stream from gpos (a multiple of 16) on: a.cap bytes, zeroed
    __shared__ uint32_t s4[4];
    const uint32_t tid = threadIdx.x;
    const uint64_t chunk = blockIdx.x;
    const int D = a.D;
    const int cpl = (D + 255) / 256;
    const int col0 = (int)tid * cpl;
    const uint64_t first = chunk * (uint64_t)a.chunk_len;
    const uint32_t n = (uint32_t)((a.total_len - first < a.chunk_len) ? (a.total_len - first) : a.chunk_len);
    const U* const sc = (const U*)a.src + first;
    uint8_t* const gdst = a.slots + chunk * a.slot_stride;
    const uint32_t cap = a.cap;
    const uint32_t win_a = lds_addr(win);

    for (uint32_t u = tid; u < (cap >> 4); u += 256u) ((uint4*)win)[u] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    uint32_t wl = a.write_size ? 8u : 0u;        // write position inside the window (bytes)
    uint32_t gpos = 0;                           // stream offset of the window's first byte
    // whole 16-byte pieces below `upto` (a multiple of 16) leave for the slot; what stays moves to the front
    auto drain = [&](uint32_t upto) {
        __syncthreads();
        for (uint32_t u = tid * 16u; u < upto; u += 256u * 16u) {
            uint4* const r = (uint4*)(win + u);
            *(uint4*)(gdst + gpos + u) = *r;
            *r = make_uint4(0, 0, 0, 0);
        }
        __syncthreads();
        // (every caller passes wl rounded down or up to 16: what stays is LESS than one piece, and nothing has been written above wl)
        if (upto != 0 && wl > upto && tid == 0) {
            const uint4 v = *(uint4*)(win + upto);
            *(uint4*)(win + upto) = make_uint4(0, 0, 0, 0);
            *(uint4*)win = v;
        }
        gpos += upto;
        wl -= upto;
        __syncthreads();
    };
    auto or_bits = [&](uint32_t bp, uint32_t v, uint32_t nb) {           // the low nb (<= 16) bits of v at window bit position bp
        if (nb == 0) return;
        const uint64_t x = (uint64_t)v << (bp & 31u);
        lds_word* q = (lds_word*)(uintptr_t)(win_a + ((bp >> 3) & ~3u));
        __hip_atomic_fetch_or(q, (uint32_t)x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((uint32_t)(x >> 32)) __hip_atomic_fetch_or(q + 1, (uint32_t)(x >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto put_run = [&](uint32_t run) {           // :377-384
        if (tid == 0) {
            win[wl] = (uint8_t)((run & 0x7fu) | (run > 0x7fu ? 0x80u : 0u));
            if (run > 0x7fu) win[wl + 1] = (uint8_t)(run >> 7);
        }
        wl += run > 0x7fu ? 2u : 1u;
    };

    const uint32_t hdr_bytes = (2u * (uint32_t)D * HB + 7u) >> 3;
    const uint32_t blk = 8u * (uint32_t)D;
    const int64_t limit = (int64_t)n - 2 * (int64_t)blk;                 // last_full_group_start (:158)
    int64_t pos_in = 0;
    uint32_t ngroups = 0, run = 0, hdr_pos = 0;
    int slot = 0;
    uint32_t pv[CPL];
    int pd[CPL], ctr[CPL];
#pragma unroll
    for (int k = 0; k < CPL; k++) { pv[k] = 0; pd[k] = 0; ctr[k] = 0; }
    auto start_group = [&]() {
        ngroups++;
        drain(wl & ~15u);
        hdr_pos = wl;
        wl += hdr_bytes;
        slot = 0;
    };
    bool active = n >= 128u && limit >= 0;       // :116 and the loop guard :160
    if (active) start_group();

    while (active) {
        // ---- the block at pos_in: forecast, zigzag, widths (:197-298)
        // (round 5 staged the block's 8 rows through LDS with 16-byte requests instead of one 2-byte load a sample: 0.75 against 0.69 ms at
        //  1 000 columns -- the image costs a resident workgroup a CU and the loads were not what the chunk's 32 serial blocks wait for)
        uint32_t z[8][CPL], nb[CPL];
        uint32_t lane_bits = 0;
#pragma unroll
        for (int k = 0; k < CPL; k++) {
            const int col = col0 + k;
            nb[k] = 0;
            if (k >= cpl || col >= D) {
#pragma unroll
                for (int i = 0; i < 8; i++) z[i][k] = 0;
                continue;
            }
            const int coef = FIRE ? fire_coef<W, false>(ctr[k]) : 0;
            int grad = 0;
            uint32_t mask = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint32_t x = (uint32_t)sc[pos_in + (int64_t)i * D + col];
                const int delta = sext<W>((int)(x - pv[k]));
                const int pred = FIRE ? fire_predict<W, false>(pd[k], coef) : 0;
                const int err = sext<W>(delta - pred);
                if (FIRE && (i & 1)) grad += sign_times(err, pd[k]);
                z[i][k] = zigzag<W>(err);
                mask |= z[i][k];
                pv[k] = x;
                pd[k] = delta;
            }
            if (FIRE) ctr[k] = wrap_counter<W>(ctr[k] + (sext<W>(grad) >> 2));
            nb[k] = nbits_of<W, false>(mask);
            lane_bits += nb[k];
        }
        uint32_t total;
        const uint32_t excl = wg_scan(lane_bits, total, s4);

        // ---- RLE state machine (:350-456); the same in every lane
        for (;;) {
            if (total == 0 && run < 0x7fffu) {
                run++;
                pos_in += blk;
                const bool more = TAIL_LE ? (pos_in <= limit) : (pos_in < limit);
                if (more) break;
                slot++;
                __syncthreads();
                put_run(run);
                wl += (uint32_t)(2 - slot);          // one 0x00 per slot the group still has (:386-391)
                run = 0;
                active = false;
                break;
            }
            if (run > 0) {
                slot++;
                __syncthreads();
                put_run(run);
                run = 0;
                if (slot == 2) start_group();        // :430-450
                continue;
            }
            const uint32_t row_bits = ((total + 7u) >> 3) << 3;
            uint32_t off = excl;
#pragma unroll
            for (int k = 0; k < CPL; k++) {
                const int col = col0 + k;
                if (k < cpl && col < D) {
                    or_bits(hdr_pos * 8u + (uint32_t)(slot * D + col) * HB, nb[k] == (uint32_t)W ? (uint32_t)(W - 1) : nb[k], HB);      // :296
#pragma unroll
                    for (int i = 0; i < 8; i++) or_bits(wl * 8u + (uint32_t)i * row_bits + off, z[i][k], nb[k]);
                    off += nb[k];
                }
            }
            wl += row_bits;                          // 8 rows of row_bits / 8 bytes
            pos_in += blk;
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
            for (uint32_t j = tid; j < m; j += 256u) win[wl + j] = tp[j];
            wl += m;
            tp += m;
            left -= m;
        }
    }
    const uint32_t total_bytes = gpos + wl;
    drain((wl + 15u) & ~15u);
    if (tid == 0) {                                  // format.h:36-45 (the window's first pieces have left: straight to the slot)
        if (a.write_size) {
            ((uint32_t*)gdst)[0] = ngroups;
            ((uint32_t*)gdst)[1] = (remaining & 0xffffu) | ((uint32_t)D << 16);
        }
        a.sizes[chunk] = total_bytes;
        if (a.rets) a.rets[chunk] = (int64_t)(total_bytes / ESZ);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// 2 048 .. 65 535 columns ("big"; the header's ndims is a full uint16, format.h:36-45): the same one-workgroup-per-chunk scheme with the
// columns taken in TILES of 2 048 (lane t: columns 2048 tile + 8 t .. + 7).  What a lane kept in registers per column cannot be kept for
// 65 535 of them: a column's last value and last delta are read back from the two rows in front of the block (decoder: its own output;
// encoder: the input), the FIRE counters live in a scratch array (counters[chunk][D], the launcher's).  A stream group is up to 2 MB here,
// so the encoder has no LDS window: its slot is zeroed first and the fields are OR-ed straight into it (global atomics); it forecasts a
// block twice -- once for the slot's total (the RLE decision and the row stride need it before any field can be placed), once to place.
// Written for completeness (a block of 65 535 uint16 columns is a megabyte; the reference's own tests stop at 129 columns), not speed.
constexpr uint32_t kBigTile = 256u * kAnyCpl;

__device__ __forceinline__ uint32_t width_of_field(uint32_t f, int W) { return f == (uint32_t)(W - 1) ? (uint32_t)W : f; }

template <int W, bool FIRE>
__global__ void __launch_bounds__(256) decode_big_kernel(DecodeArgs a, int32_t* counters)
{
    using U = typename Elem<W>::U;
    constexpr int HB = Elem<W>::HB;
    constexpr uint32_t MASK = Elem<W>::MASK;
    constexpr int ESZ = W / 8;
    __shared__ uint32_t s4[4];
    const uint32_t tid = threadIdx.x;
    const uint64_t chunk = blockIdx.x;
    const uint32_t D = (uint32_t)a.D;
    const uint32_t ntile = (D + kBigTile - 1) / kBigTile;
    const uint64_t off_c = a.offsets[chunk];
    const uint8_t* const s = a.comp + off_c;
    const uint64_t slen64 = a.offsets[chunk + 1] - off_c;
    const uint32_t stream_len = slen64 < 0xffffffffull ? (uint32_t)slen64 : 0xffffffffu;
    U* const o = (U*)a.out + chunk * (uint64_t)a.chunk_len;
    int32_t* const ctr = FIRE ? counters + chunk * (uint64_t)D : nullptr;
    if constexpr (FIRE)
        for (uint32_t c = tid; c < D; c += 256u) ctr[c] = 0;          // (zeroed by whoever; read back only after the barrier below)
    __syncthreads();

    uint32_t groups_left, remaining, pos;
    bool corrupt = false;
    if (!a.noheader) {
        if (stream_len < 8u) { corrupt = true; groups_left = 0; remaining = 0; pos = 0; }
        else {
            const uint32_t w0 = load_u32_any(s), w1 = load_u32_any(s + 4);
            groups_left = w0;
            remaining = w1 & 0xffffu;
            pos = 8;
            if ((w1 >> 16) != D) corrupt = true;
        }
    } else {
        groups_left = a.nh_ngroups;
        remaining = a.nh_remaining;
        pos = 0;
    }
    const uint32_t hdr_bytes = (2u * D * HB + 7u) >> 3;
    const uint32_t blk_elems = 8u * D;
    if (groups_left > a.chunk_len / blk_elems + 2u) corrupt = true;      // a damaged header must not make the loop spin
    if (corrupt) groups_left = 0;
    uint32_t out_elems = 0;

    // one block: slot `slot` of the group whose header sits at s + hdr_at; payload rows at s + pay (packed blocks), zero errors in a run
    auto emit_block = [&](bool run_block, uint32_t hdr_at, int slot, uint32_t pay, uint32_t row_bits) {
        U* const ob = o + out_elems;
        uint32_t carry = 0;
        for (uint32_t tile = 0; tile < ntile; tile++) {
            const uint32_t col0 = tile * kBigTile + tid * kAnyCpl;
            uint32_t nb[kAnyCpl], lane_bits = 0;
#pragma unroll
            for (int k = 0; k < kAnyCpl; k++) {
                nb[k] = 0;
                if (!run_block && col0 + k < D) nb[k] = width_of_field(fetch_bits(s + hdr_at, ((uint32_t)slot * D + col0 + k) * HB, HB), W);
                lane_bits += nb[k];
            }
            uint32_t off = 0;
            if (!run_block) {                                         // (uniform)
                uint32_t tile_total;
                off = carry + wg_scan(lane_bits, tile_total, s4);
                carry += tile_total;
            }
#pragma unroll
            for (int k = 0; k < kAnyCpl; k++) {
                const uint32_t col = col0 + k;
                if (col >= D) continue;
                uint32_t pv = 0;
                int pd = 0;
                if (out_elems) {                                      // the two rows in front (a block is 8 rows: both exist)
                    pv = (uint32_t)o[out_elems - D + col];
                    pd = sext<W>((int)(pv - (uint32_t)o[out_elems - 2u * D + col]));
                }
                int c = FIRE ? ctr[col] : 0;
                int coef = FIRE ? fire_coef<W, false>(c) : 0;
                if constexpr (FIRE && W == 16) {
                    if (a.quirk && run_block) coef = fire_coef_ref_run16(c, (int)col);
                }
                int grad = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const uint32_t z = run_block ? 0u : fetch_bits(s + pay, (uint32_t)i * row_bits + off, nb[k]);
                    const int err = unzigzag(z);
                    const int pred = FIRE ? fire_predict<W, false>(pd, coef) : 0;
                    const int delta = sext<W>(err + pred);
                    if (FIRE && (i & 1)) grad += sign_times(err, pd);
                    pv = (pv + (uint32_t)delta) & MASK;
                    pd = delta;
                    ob[(uint32_t)i * D + col] = (U)pv;
                }
                if constexpr (FIRE) ctr[col] = wrap_counter<W>(c + (sext<W>(grad) >> 2));
                off += nb[k];
            }
        }
        out_elems += blk_elems;
    };

    while (groups_left > 0 && !corrupt) {
        groups_left--;
        if (hdr_bytes > stream_len - pos) { corrupt = true; break; }
        const uint32_t hdr_at = pos;
        // the slots' totals: every lane its columns of every tile, one reduction
        uint32_t both_lo = 0, both_hi = 0;                                // (a slot's total is at most 65 535 * 16 < 2^21)
        for (uint32_t tile = 0; tile < ntile; tile++)
#pragma unroll
            for (int k = 0; k < kAnyCpl; k++) {
                const uint32_t col = tile * kBigTile + tid * kAnyCpl + k;
                if (col < D) {
                    both_lo += width_of_field(fetch_bits(s + hdr_at, col * HB, HB), W);
                    both_hi += width_of_field(fetch_bits(s + hdr_at, (D + col) * HB, HB), W);
                }
            }
        uint32_t tot0, tot1;
        (void)wg_scan(both_lo, tot0, s4);
        (void)wg_scan(both_hi, tot1, s4);
        pos += hdr_bytes;
        for (int slot = 0; slot < 2 && !corrupt; slot++) {
            const uint32_t total = slot ? tot1 : tot0;
            if (total == 0) {                                             // RUN slot: varint length in blocks (:829-833)
                if (stream_len - pos < 2u) {
                    if (stream_len == pos || (load_u8(s + pos) & 0x80u)) { corrupt = true; break; }
                }
                const uint32_t b0 = load_u8(s + pos);
                uint32_t len = b0 & 0x7fu;
                if (b0 & 0x80u) { len |= load_u8(s + pos + 1) << 7; pos += 2; }
                else pos += 1;
                for (; len > 0; len--) {
                    if (out_elems + blk_elems > a.chunk_len) { corrupt = true; break; }
                    emit_block(true, hdr_at, slot, 0, 0);
                }
            } else {                                                      // packed block: 8 rows of ceil(total / 8) bytes (:961-990)
                const uint32_t row_bits = ((total + 7u) >> 3) << 3;
                if (row_bits > stream_len - pos || out_elems + blk_elems > a.chunk_len) { corrupt = true; break; }
                emit_block(false, hdr_at, slot, pos, row_bits);
                pos += row_bits;
            }
        }
    }

    // ---- verbatim tail (:1171)
    if (!corrupt && (out_elems + remaining > a.chunk_len || (uint64_t)remaining * ESZ > (uint64_t)(stream_len - pos))) corrupt = true;
    if (!corrupt) {
        const uint8_t* const t = s + pos;
        uint8_t* const d = (uint8_t*)(o + out_elems);
        for (uint32_t j = tid; j < remaining * ESZ; j += 256u) d[j] = t[j];
    }
    if (tid == 0 && a.rets) a.rets[chunk] = corrupt ? kErrCorrupt : (int64_t)out_elems + remaining;
}

template <int W, bool FIRE>
__global__ void __launch_bounds__(256) encode_big_kernel(EncodeArgs a, int32_t* counters)
{
    using U = typename Elem<W>::U;
    constexpr int HB = Elem<W>::HB;
    constexpr int ESZ = W / 8;
    constexpr bool TAIL_LE = FIRE;               // "<=" at sprintz_xff_rle.cpp:362, "<" at sprintz_delta_rle.cpp:226
    __shared__ uint32_t s4[4];
    const uint32_t tid = threadIdx.x;
    const uint64_t chunk = blockIdx.x;
    const uint32_t D = (uint32_t)a.D;
    const uint32_t ntile = (D + kBigTile - 1) / kBigTile;
    const uint64_t first = chunk * (uint64_t)a.chunk_len;
    const uint32_t n = (uint32_t)((a.total_len - first < a.chunk_len) ? (a.total_len - first) : a.chunk_len);
    const U* const sc = (const U*)a.src + first;
    uint8_t* const gdst = a.slots + chunk * a.slot_stride;
    int32_t* const ctr = FIRE ? counters + chunk * (uint64_t)D : nullptr;

    for (uint64_t u = (uint64_t)tid * 16u; u + 16u <= a.slot_stride; u += 256u * 16u) *(uint4*)(gdst + u) = make_uint4(0, 0, 0, 0);
    if constexpr (FIRE)
        for (uint32_t c = tid; c < D; c += 256u) ctr[c] = 0;
    __syncthreads();
    uint32_t wl = a.write_size ? 8u : 0u;        // write position in the slot (bytes)
    auto or_bits = [&](uint64_t bp, uint32_t v, uint32_t nb) {            // the low nb (<= 16) bits of v at slot bit position bp
        if (nb == 0 || v == 0) return;
        const uint64_t x = (uint64_t)v << (bp & 31u);
        uint32_t* const q = (uint32_t*)(gdst + ((bp >> 3) & ~(uint64_t)3));
        __hip_atomic_fetch_or(q, (uint32_t)x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(x >> 32)) __hip_atomic_fetch_or(q + 1, (uint32_t)(x >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto put_run = [&](uint32_t run) {           // :377-384 (the slot is zero there: plain byte stores)
        if (tid == 0) {
            gdst[wl] = (uint8_t)((run & 0x7fu) | (run > 0x7fu ? 0x80u : 0u));
            if (run > 0x7fu) gdst[wl + 1] = (uint8_t)(run >> 7);
        }
        wl += run > 0x7fu ? 2u : 1u;
    };
    const uint32_t hdr_bytes = (2u * D * HB + 7u) >> 3;
    const uint32_t blk = 8u * D;
    const int64_t limit = (int64_t)n - 2 * (int64_t)blk;                 // last_full_group_start (:158)
    int64_t pos_in = 0;
    uint32_t ngroups = 0, run = 0, hdr_pos = 0;
    int slot = 0;
    auto start_group = [&]() { ngroups++; hdr_pos = wl; wl += hdr_bytes; slot = 0; };
    // column col of the block at pos_in: forecast, zigzag, width (:197-298); c: its counter, updated
    auto forecast = [&](uint32_t col, uint32_t (&z)[8], int& c) -> uint32_t {
        uint32_t pv = 0;
        int pd = 0;
        if (pos_in) {                                                     // the two rows in front (a block is 8 rows: both exist)
            pv = (uint32_t)sc[pos_in - D + col];
            pd = sext<W>((int)(pv - (uint32_t)sc[pos_in - 2 * (int64_t)D + col]));
        }
        const int coef = FIRE ? fire_coef<W, false>(c) : 0;
        int grad = 0;
        uint32_t mask = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t x = (uint32_t)sc[pos_in + (int64_t)i * D + col];
            const int delta = sext<W>((int)(x - pv));
            const int pred = FIRE ? fire_predict<W, false>(pd, coef) : 0;
            const int err = sext<W>(delta - pred);
            if (FIRE && (i & 1)) grad += sign_times(err, pd);
            z[i] = zigzag<W>(err);
            mask |= z[i];
            pv = x;
            pd = delta;
        }
        if (FIRE) c = wrap_counter<W>(c + (sext<W>(grad) >> 2));
        return nbits_of<W, false>(mask);
    };
    bool active = n >= 128u && limit >= 0;       // :116 and the loop guard :160
    if (active) start_group();

    while (active) {
        // ---- the block's total width (pass A: nothing kept, no counter moved)
        uint32_t lane_total = 0;
        for (uint32_t tile = 0; tile < ntile; tile++)
#pragma unroll
            for (int k = 0; k < kAnyCpl; k++) {
                const uint32_t col = tile * kBigTile + tid * kAnyCpl + k;
                if (col < D) { uint32_t z[8]; int c = FIRE ? ctr[col] : 0; lane_total += forecast(col, z, c); }
            }
        uint32_t total;
        (void)wg_scan(lane_total, total, s4);

        // ---- RLE state machine (:350-456); the same in every lane.  (A block of zero errors moves no counter: runs need no second pass.)
        for (;;) {
            if (total == 0 && run < 0x7fffu) {
                run++;
                pos_in += blk;
                const bool more = TAIL_LE ? (pos_in <= limit) : (pos_in < limit);
                if (more) break;
                slot++;
                put_run(run);
                wl += (uint32_t)(2 - slot);          // one 0x00 per slot the group still has (:386-391)
                run = 0;
                active = false;
                break;
            }
            if (run > 0) {
                slot++;
                put_run(run);
                run = 0;
                if (slot == 2) start_group();        // :430-450
                continue;
            }
            // ---- pass B: the fields into the header, the rows into the payload, the counters on
            const uint32_t row_bits = ((total + 7u) >> 3) << 3;
            uint32_t carry = 0;
            for (uint32_t tile = 0; tile < ntile; tile++) {
                uint32_t z[kAnyCpl][8], nb[kAnyCpl], lane_bits = 0;
#pragma unroll
                for (int k = 0; k < kAnyCpl; k++) {
                    const uint32_t col = tile * kBigTile + tid * kAnyCpl + k;
                    nb[k] = 0;
                    if (col < D) {
                        int c = FIRE ? ctr[col] : 0;
                        nb[k] = forecast(col, z[k], c);
                        if constexpr (FIRE) ctr[col] = c;
                    }
                    lane_bits += nb[k];
                }
                uint32_t tile_total;
                uint32_t off = carry + wg_scan(lane_bits, tile_total, s4);
                carry += tile_total;
#pragma unroll
                for (int k = 0; k < kAnyCpl; k++) {
                    const uint32_t col = tile * kBigTile + tid * kAnyCpl + k;
                    if (col < D) {
                        or_bits((uint64_t)hdr_pos * 8u + ((uint64_t)slot * D + col) * HB, nb[k] == (uint32_t)W ? (uint32_t)(W - 1) : nb[k], HB);      // :296
#pragma unroll
                        for (int i = 0; i < 8; i++) or_bits((uint64_t)wl * 8u + (uint64_t)i * row_bits + off, z[k][i], nb[k]);
                        off += nb[k];
                    }
                }
            }
            wl += row_bits;                          // 8 rows of row_bits / 8 bytes
            pos_in += blk;
            slot++;
            if (slot == 2) {
                if (pos_in <= limit) start_group();
                else active = false;
            }
            break;
        }
    }

    // ---- verbatim tail (:553): the slot is zero there
    const uint32_t remaining = (uint32_t)((int64_t)n - pos_in);
    {
        const uint8_t* const tp = (const uint8_t*)(sc + pos_in);
        for (uint32_t j = tid; j < remaining * ESZ; j += 256u) gdst[wl + j] = tp[j];
    }
    const uint32_t total_bytes = wl + remaining * ESZ;
    if (tid == 0) {                                  // format.h:36-45
        if (a.write_size) {
            ((uint32_t*)gdst)[0] = ngroups;
            ((uint32_t*)gdst)[1] = (remaining & 0xffffu) | (D << 16);
        }
        a.sizes[chunk] = total_bytes;
        if (a.rets) a.rets[chunk] = (int64_t)(total_bytes / ESZ);
    }
}

template <typename K, typename A>
hipError_t launch_any(K kernel, unsigned grid, size_t shmem, hipStream_t st, const A& a)
{
    return launch_with_lds(kernel, grid, 256u, shmem, st, a);      // (> 48 KB: the attribute once per instantiation and device, lds_attr.h)
}

}  // namespace

hipError_t launch_decode_any(int w, bool fire, unsigned grid, hipStream_t st, const DecodeArgs& a)
{
    // [header image | payload image | block image] (decode_any_kernel): <= 2 KB + 2 x 32 KB at 2 047 uint16 columns
    const uint32_t D = (uint32_t)a.D, esz = (uint32_t)w / 8, hb = w == 8 ? 3u : 4u;
    const size_t shmem = (size_t)(((((2u * D * hb + 7u) >> 3) + 15u) & ~15u) + 32u) + 2u * (size_t)(((8u * D * esz + 15u) & ~15u) + 32u);
    // (columns a lane: ceil(D / 256); the kernels are built for <= 4 and <= 8 -- at <= 1 024 columns half the registers: 4 workgroups a CU instead of 3)
    if (D <= 1024u) {
        if (w == 8) return fire ? launch_any(decode_any_kernel<8, true, 4>, grid, shmem, st, a) : launch_any(decode_any_kernel<8, false, 4>, grid, shmem, st, a);
        return fire ? launch_any(decode_any_kernel<16, true, 4>, grid, shmem, st, a) : launch_any(decode_any_kernel<16, false, 4>, grid, shmem, st, a);
    }
    if (w == 8) return fire ? launch_any(decode_any_kernel<8, true, 8>, grid, shmem, st, a) : launch_any(decode_any_kernel<8, false, 8>, grid, shmem, st, a);
    return fire ? launch_any(decode_any_kernel<16, true, 8>, grid, shmem, st, a) : launch_any(decode_any_kernel<16, false, 8>, grid, shmem, st, a);
}
// 2 048 .. 65 535 columns; counters: nchunks * ndims int32 of scratch (FIRE codecs only; may be null otherwise)
hipError_t launch_decode_big(int w, bool fire, unsigned grid, hipStream_t st, const DecodeArgs& a, int32_t* counters)
{
    if (w == 8) { if (fire) hipLaunchKernelGGL((decode_big_kernel<8, true>), dim3(grid), dim3(256), 0, st, a, counters); else hipLaunchKernelGGL((decode_big_kernel<8, false>), dim3(grid), dim3(256), 0, st, a, counters); }
    else { if (fire) hipLaunchKernelGGL((decode_big_kernel<16, true>), dim3(grid), dim3(256), 0, st, a, counters); else hipLaunchKernelGGL((decode_big_kernel<16, false>), dim3(grid), dim3(256), 0, st, a, counters); }
    return hipGetLastError();
}
hipError_t launch_encode_big(int w, bool fire, unsigned grid, hipStream_t st, const EncodeArgs& a, int32_t* counters)
{
    if (w == 8) { if (fire) hipLaunchKernelGGL((encode_big_kernel<8, true>), dim3(grid), dim3(256), 0, st, a, counters); else hipLaunchKernelGGL((encode_big_kernel<8, false>), dim3(grid), dim3(256), 0, st, a, counters); }
    else { if (fire) hipLaunchKernelGGL((encode_big_kernel<16, true>), dim3(grid), dim3(256), 0, st, a, counters); else hipLaunchKernelGGL((encode_big_kernel<16, false>), dim3(grid), dim3(256), 0, st, a, counters); }
    return hipGetLastError();
}
hipError_t launch_encode_any(int w, bool fire, unsigned grid, size_t shmem, hipStream_t st, const EncodeArgs& a)
{
    if (a.D <= 1024) {
        if (w == 8) return fire ? launch_any(encode_any_kernel<8, true, 4>, grid, shmem, st, a) : launch_any(encode_any_kernel<8, false, 4>, grid, shmem, st, a);
        return fire ? launch_any(encode_any_kernel<16, true, 4>, grid, shmem, st, a) : launch_any(encode_any_kernel<16, false, 4>, grid, shmem, st, a);
    }
    if (w == 8) return fire ? launch_any(encode_any_kernel<8, true, 8>, grid, shmem, st, a) : launch_any(encode_any_kernel<8, false, 8>, grid, shmem, st, a);
    return fire ? launch_any(encode_any_kernel<16, true, 8>, grid, shmem, st, a) : launch_any(encode_any_kernel<16, false, 8>, grid, shmem, st, a);
}

}  // namespace sprintz

// encode_kernel.h -- batched Sprintz encoder for gfx950, generic lane mapping.
//
// Replaces (bit-exact: stream bytes, byte length, element-count return value):
//   compress_rowmajor_xff_rle<>              sprintz_xff_rle.cpp:61-555
//   compress_rowmajor_delta_rle<>            sprintz_delta_rle.cpp:55-404
//   compress_rowmajor_{delta,xff}_rle_lowdim<>   sprintz_delta_lowdim.cpp:39-384,
//                                                sprintz_xff_lowdim.cpp:44-400
//
// Output path: every group owns a zero-initialised LDS ring of `cap` bytes (a
// sliding window over its chunk's output stream).  Lanes OR their bit fields
// into it with ds_or_b32 (the reference ORs pext results into zeroed memory,
// sprintz_xff_rle.cpp:483-517); whenever a new stream group starts, the
// 16-byte-aligned prefix of the window is flushed to HBM with dwordx4 stores
// and re-zeroed.  No byte-granular HBM stores anywhere.
#pragma once

#include "sprintz_device.h"
#include "compact_tail.h"

namespace sprintz {

struct EncodeArgs {
    const void* src;            // total_len elements, row-major
    uint64_t total_len;
    uint32_t chunk_len;
    uint64_t nchunks;
    int D;
    int log2DP;
    uint8_t* slots;             // chunk c's stream at slots + c*slot_stride (16-byte aligned)
    uint64_t slot_stride;
    uint32_t* sizes;            // exact stream bytes per chunk
    int64_t* rets;              // optional: reference's element-count return value
    int write_size;             // 0: omit the 8-byte header (sprintz_xff_rle.cpp:119-127)
    uint32_t cap;               // ring bytes per group (power of two, >= max group bytes + 32)
    uint32_t lds_group_stride;  // encode_fast: LDS bytes per group (ring + input staging)
    // column-major source (BASELINE config 5): element (row r, column d) at src[d*col_stride + r];
    // chunk c covers rows [c*chunk_len/D, ...) and is coded as the reference codes the row-major
    // flattening of that row range.  0 = row-major.
    uint64_t col_stride;
    // non-RLE codecs (generic kernel only), see DecodeArgs
    int norle;
    int raw;
    // the dense container written by the encode launch itself (compact_tail.h; encode_fast only): dn.dense == null -> slots only
    DenseArgs dn;
    // a single call on the caller thread's mapped host buffer (encode_lat.h alone): the kernel ends by writing host_ticket to
    // host_flag, after every lane's stores
    uint64_t* host_flag;
    uint64_t host_ticket;
};

template <int W, bool FIRE, bool LOWDIM, int CPL>
__global__ void __launch_bounds__(kThreads) encode_kernel(EncodeArgs a)
{
    using U = typename Elem<W>::U;
    constexpr int HB = Elem<W>::HB;
    constexpr int ESZ = W / 8;
    constexpr bool TAIL_LE = FIRE && !LOWDIM;   // "<=" at sprintz_xff_rle.cpp:362, "<" in the other three codecs
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int DP = 1 << a.log2DP;
    const int D = a.D;
    const uint64_t gtid = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
    const uint64_t chunk = gtid >> a.log2DP;
    const int lane_d = (int)(threadIdx.x & (uint32_t)(DP - 1));
    if (chunk >= a.nchunks) return;

    const uint64_t first = chunk * (uint64_t)a.chunk_len;
    const uint32_t n = (uint32_t)((a.total_len - first < a.chunk_len) ? (a.total_len - first) : a.chunk_len);
    const U* const sc = (const U*)a.src + first;
    uint8_t* const gdst = a.slots + chunk * a.slot_stride;
    const uint64_t cs = a.col_stride;
    const U* const cm0 = (const U*)a.src + (cs ? first / (uint64_t)D : 0);     // column 0 at this chunk's first row
    // element e of the chunk in row-major order
    auto elem = [&](uint32_t e) -> uint32_t {
        return cs ? (uint32_t)cm0[(uint64_t)(e % (uint32_t)D) * cs + e / (uint32_t)D] : (uint32_t)sc[e];
    };

    // A chunk too short for one group (n < 128 or n < 16 D: BASELINE config 3 at 1 KB chunks) is
    // stored verbatim behind its header (sprintz_xff_rle.cpp:116-124, :158-160): straight copy,
    // 16 bytes per lane, instead of the byte-wise trip through the LDS ring below.
    if (!a.norle && !cs && !(n >= 128u && (int64_t)n - 16 * (int64_t)D >= 0)) {
        const uint32_t hdr = a.write_size ? 8u : 0u;
        const uint8_t* src = (const uint8_t*)sc;
        uint8_t* dst = gdst + hdr;
        const uint32_t nbytes = n * ESZ;
        copy_verbatim<false>(src, dst, nbytes, (uint32_t)lane_d, (uint32_t)DP);      // (the compaction pass reads it next: ordinary stores)
        // slots are zero padded to 16 bytes (sprintz_mi355x_compact copies whole 16-byte units: the container's
        // alignment padding must not carry stale workspace bytes)
        for (uint32_t j = hdr + nbytes + (uint32_t)lane_d; j < ((hdr + nbytes + 15u) & ~15u); j += (uint32_t)DP) gdst[j] = 0;
        if (lane_d == 0) {
            if (a.write_size) {
                ((uint32_t*)gdst)[0] = 0;
                ((uint32_t*)gdst)[1] = (n & 0xffffu) | ((uint32_t)D << 16);
            }
            a.sizes[chunk] = hdr + nbytes;
            if (a.rets) a.rets[chunk] = (int64_t)((hdr + nbytes) / ESZ);
        }
        return;
    }

    const uint32_t cap = a.cap, capm = cap - 1;
    // +16 per group: with few lanes per group (univariate: one) a power-of-two stride puts every
    // group's ring on the same banks
    uint8_t* const ring = smem + (size_t)(threadIdx.x >> a.log2DP) * (cap + 16u);
    uint32_t* const ring32 = (uint32_t*)ring;

    // zero this group's ring
    for (uint32_t u = (uint32_t)lane_d; u < (cap >> 4); u += (uint32_t)DP) ((uint4*)ring)[u] = make_uint4(0, 0, 0, 0);
    wave_lds_sync();

    uint32_t wpos = a.norle == 1 ? 6u : ((a.norle == 2 || a.write_size) ? 8u : 0u);   // stream write position (bytes)
    uint32_t flushed = 0;                     // multiple of 16; ring holds [flushed, flushed + cap)

    // flush [flushed, upto) (upto multiple of 16) to HBM and re-zero it
    auto flush_to = [&](uint32_t upto) {
        wave_lds_sync();
        const uint32_t nunits = (upto - flushed) >> 4;
        for (uint32_t u = (uint32_t)lane_d; u < nunits; u += (uint32_t)DP) {
            const uint32_t p = flushed + (u << 4);
            uint4* r = (uint4*)(ring + (p & capm));
            *(uint4*)(gdst + p) = *r;
            *r = make_uint4(0, 0, 0, 0);
        }
        flushed = upto;
        wave_lds_sync();
    };
    // OR the low nb (<= 16) bits of v at absolute stream bit position bp
    auto or_bits = [&](uint32_t bp, uint32_t v, uint32_t nb) {
        if (nb == 0) return;
        const uint32_t w = (bp >> 5), sh = bp & 31u;
        const uint32_t wm = (cap >> 2) - 1;
        atomicOr(&ring32[w & wm], v << sh);
        if (sh + nb > 32u) atomicOr(&ring32[(w + 1) & wm], v >> (32u - sh));
    };
    auto put_run = [&](uint32_t run) {          // sprintz_xff_rle.cpp:377-384
        if (lane_d == 0) {
            ring[wpos & capm] = (uint8_t)((run & 0x7fu) | (run > 0x7fu ? 0x80u : 0u));
            if (run > 0x7fu) ring[(wpos + 1) & capm] = (uint8_t)(run >> 7);
        }
        wpos += run > 0x7fu ? 2u : 1u;
    };

    const uint32_t hdr_bytes = (2u * (uint32_t)D * HB + 7u) >> 3;
    const uint32_t blk = 8u * (uint32_t)D;
    const int64_t limit = (int64_t)n - 2 * (int64_t)blk;    // last_full_group_start (:158)
    int64_t pos_in = 0;
    uint32_t ngroups = 0, run = 0, hdr_pos = 0;
    int slot = 0;

    uint32_t pv[CPL];
    int pd[CPL], ctr[CPL];
#pragma unroll
    for (int k = 0; k < CPL; k++) { pv[k] = 0; pd[k] = 0; ctr[k] = 0; }

    auto start_group = [&]() {
        ngroups++;
        flush_to(wpos & ~15u);
        hdr_pos = wpos;
        wpos += hdr_bytes;
        slot = 0;
    };

    bool active = n >= 128u && limit >= 0;      // :116 and the loop guard :160
    if (active) start_group();

    while (active) {
        // ---- forecast + zigzag + OR-mask for the block at pos_in (:197-298)
        uint32_t z[8][CPL], nb[CPL], lane_bits = 0;
#pragma unroll
        for (int k = 0; k < CPL; k++) {
            const int col = lane_d * CPL + k;
            uint32_t x[8];
            if (cs) {
                const U* const cp = cm0 + (uint64_t)col * cs + (uint32_t)pos_in / (uint32_t)D;
#pragma unroll
                for (int i = 0; i < 8; i++) x[i] = (col < D) ? (uint32_t)cp[i] : 0u;
            } else if (LOWDIM && D == 1) {
                // one column: the block's 8 samples are contiguous -- one load, not eight
                typedef uint32_t v2 __attribute__((ext_vector_type(2)));
                typedef uint32_t v4 __attribute__((ext_vector_type(4)));
                typedef v2 __attribute__((aligned(1), may_alias)) v2u;
                typedef v4 __attribute__((aligned(1), may_alias)) v4u;
                if constexpr (W == 8) {
                    const v2 t = *(const v2u*)(sc + pos_in);
#pragma unroll
                    for (int i = 0; i < 8; i++) x[i] = ((i < 4 ? t.x : t.y) >> (8 * (i & 3))) & 0xffu;
                } else {
                    const v4 t = *(const v4u*)(sc + pos_in);
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const uint32_t d = (i >> 1) == 0 ? t.x : (i >> 1) == 1 ? t.y : (i >> 1) == 2 ? t.z : t.w;
                        x[i] = (i & 1) ? d >> 16 : d & 0xffffu;
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; i++) x[i] = (col < D) ? (uint32_t)sc[pos_in + (int64_t)i * D + col] : 0u;
            }
            const int coef = FIRE ? fire_coef<W, LOWDIM>(ctr[k]) : 0;
            int grad = 0;
            uint32_t mask = 0, pvk = pv[k];
            int pdk = pd[k];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int delta = sext<W>((int)(x[i] - pvk));
                const int pred = FIRE ? fire_predict<W, LOWDIM>(pdk, coef) : 0;
                const int err = sext<W>(delta - pred);
                const uint32_t zz = (!FIRE && a.raw) ? x[i] : zigzag<W>(err);   // raw: the samples themselves are packed
                if (FIRE && (i & 1)) grad += sign_times(err, pdk);
                mask |= zz;
                z[i][k] = zz;
                pvk = x[i];
                pdk = delta;
            }
            pv[k] = pvk;
            pd[k] = pdk;
            if (FIRE) ctr[k] = wrap_counter<W>(ctr[k] + (sext<W>(grad) >> 2));
            nb[k] = (col < D) ? nbits_of<W, LOWDIM>(mask) : 0u;
            lane_bits += nb[k];
        }
        uint32_t total;
        const uint32_t excl = group_excl_scan(lane_bits, lane_d, DP, total);

        // ---- RLE state machine (:350-456); everything here is group-uniform
        for (;;) {
            if (total == 0 && run < 0x7fffu && !a.norle) {
                run++;
                pos_in += blk;
                const bool more = TAIL_LE ? (pos_in <= limit) : (pos_in < limit);
                if (more) break;                         // analyse the next block
                slot++;                                  // not enough input left: close the run
                put_run(run);
                wpos += (uint32_t)(2 - slot);            // empty slots are one 0x00 byte each (ring is zero)
                run = 0;
                active = false;
                break;
            }
            if (run > 0) {                               // a run just ended
                slot++;
                put_run(run);
                run = 0;
                if (slot == 2) start_group();            // :430-450
                continue;                                // re-evaluate this block
            }
            // header fields (W -> W-1, :296) and payload
            uint32_t off = excl;
#pragma unroll
            for (int k = 0; k < CPL; k++) {
                const int col = lane_d * CPL + k;
                if (col < D) {
                    const uint32_t f = nb[k] == (uint32_t)W ? (uint32_t)(W - 1) : nb[k];
                    or_bits(hdr_pos * 8u + (uint32_t)(slot * D + col) * HB, f, HB);
                }
                if constexpr (!LOWDIM) {
                    const uint32_t row_bits = ((total + 7u) >> 3) << 3;
#pragma unroll
                    for (int i = 0; i < 8; i++) or_bits(wpos * 8u + (uint32_t)i * row_bits + off, z[i][k], nb[k]);
                } else {
#pragma unroll
                    for (int i = 0; i < 8; i++) or_bits((wpos + off) * 8u + (uint32_t)i * nb[k], z[i][k], nb[k]);
                }
                off += nb[k];
            }
            wpos += LOWDIM ? total : (((total + 7u) >> 3) << 3);
            pos_in += blk;
            slot++;
            if (slot == 2) {
                if (pos_in <= limit) start_group();
                else active = false;
            }
            break;
        }
    }

    // ---- verbatim tail through the ring (:553)
    const uint32_t remaining = (uint32_t)((int64_t)n - pos_in);
    {
        const uint8_t* tp = (const uint8_t*)(sc + pos_in);
        uint32_t left = remaining * ESZ;
        while (left > 0) {
            flush_to(wpos & ~15u);
            const uint32_t room = cap - (wpos - flushed);
            const uint32_t m = left < room ? left : room;
            if (cs) {
                const uint32_t done = remaining * ESZ - left;          // tail bytes already copied
                for (uint32_t j = (uint32_t)lane_d; j < m; j += (uint32_t)DP) {
                    const uint32_t tb = done + j;
                    ring[(wpos + j) & capm] = (uint8_t)(elem((uint32_t)pos_in + tb / ESZ) >> (8u * (tb % ESZ)));
                }
            } else {
                for (uint32_t j = (uint32_t)lane_d; j < m; j += (uint32_t)DP) ring[(wpos + j) & capm] = tp[j];
            }
            wpos += m;
            tp += m;
            left -= m;
        }
    }
    flush_to((wpos + 15u) & ~15u);

    // ---- 8-byte stream header (format.h:36-45); lane 0 also wrote unit 0 in flush_to
    if (lane_d == 0) {
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
        a.sizes[chunk] = wpos;
        if (a.rets) a.rets[chunk] = (int64_t)(wpos / ESZ);   // element units, floor (:554)
    }
}

}  // namespace sprintz

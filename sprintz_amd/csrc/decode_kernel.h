// decode_kernel.h -- batched Sprintz decoder for gfx950, generic lane mapping.
//
// Replaces (bit-exact on every stream the reference encoder emits):
//   decompress_rowmajor_xff_rle<>           sprintz_xff_rle.cpp:569-1179
//   decompress_rowmajor_delta_rle<>         sprintz_delta_rle.cpp:418-772
//   decompress_rowmajor_{delta,xff}_rle_lowdim<>  sprintz_delta_lowdim.cpp:398-794,
//                                                 sprintz_xff_lowdim.cpp:414-1119
// The 16-bit FIRE run replay implements the inverse of the reference ENCODER
// (coefficient << 12); the reference decoder's run path uses << 4
// (sprintz_xff_rle.cpp:894-901) and does not invert its own encoder there --
// see DESIGN.md "Reference decoder quirk".
#pragma once

#include "sprintz_device.h"

namespace sprintz {

struct DecodeArgs {
    const uint8_t* comp;        // compressed bytes
    const uint64_t* offsets;    // [nchunks] byte offset of each chunk stream
    uint64_t nchunks;
    uint32_t chunk_len;         // elements per decoded chunk slot (output stride)
    int D;                      // ndims
    int log2DP;                 // lanes per chunk = 1 << log2DP
    void* out;                  // decoded elements, chunk c at out + c*chunk_len
    int64_t* rets;              // optional per-chunk element counts
    int vec_store;              // 1: LDS-transposed 16-byte stores are legal (alignment checked on host)
    uint32_t lds_group_stride;  // bytes of LDS per group when vec_store
    // headerless form (sprintz_xff.h:56-58)
    int noheader;
    uint32_t nh_ngroups;
    uint32_t nh_remaining;
    uint32_t chunks_per_group;  // decode_fast: consecutive chunks decoded by one lane group
    // query-on-compressed (sprintz_delta.h:95-98, sprintz_xff.h:90-93, query.hpp:23-29): kernels
    // instantiated with Q != 0 reduce every column of every chunk while decoding
    // column-major destination (BASELINE config 5): element (row r, column d) at out[d*col_stride + r];
    // chunk c holds rows [c*chunk_len/D, ...).  0 = row-major.
    uint64_t col_stride;
    // non-RLE codecs (sprintz_delta.cpp:64-1391; generic kernel only): 6-byte header {u32 len; u16 ndims},
    // len/(16 D) groups, an all-zero block has no payload and no run length; raw: bit-packing only
    int norle;
    int raw;
    int quirk;                  // 1: replay the runs of 16-bit general-layout FIRE streams as the REFERENCE DECODER does (fire_coef_ref_run16)
    int qop;                    // 1: max, 2: sum (what lands in qres)
    uint64_t* qres;             // [nchunks][D] per-chunk, per-column partial results
    // a single call on the caller thread's mapped host buffer (decode_lat.h alone): offsets == null -> the one chunk's stream is
    // comp[one_off0, one_off1); host_flag != null -> the kernel ends by writing host_ticket there, after every lane's stores
    uint64_t one_off0, one_off1;
    uint64_t* host_flag;
    uint64_t host_ticket;
};

// Q (template): 0 = plain decode; 1 = decode + reduce; 2 = reduce only (nothing is written
// to `out` -- QueryParams::materialize == false)
constexpr int kQueryOff = 0, kQueryMaterialize = 1, kQueryReduceOnly = 2;

constexpr int64_t kErrCorrupt = -5;

template <int W, bool FIRE, bool LOWDIM, int CPL, int Q = 0>
__global__ void __launch_bounds__(kThreads) decode_kernel(DecodeArgs a)
{
    using U = typename Elem<W>::U;
    constexpr int HB = Elem<W>::HB;
    constexpr uint32_t MASK = Elem<W>::MASK;
    constexpr int ESZ = W / 8;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int DP = 1 << a.log2DP;
    const int D = a.D;
    const uint64_t gtid = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
    const uint64_t chunk = gtid >> a.log2DP;
    const int lane_d = (int)(threadIdx.x & (uint32_t)(DP - 1));
    if (chunk >= a.nchunks) return;                 // whole groups leave together

    const uint64_t off_c = a.offsets[chunk];
    const uint8_t* s = a.comp + off_c;
    // the stream ends where the next one starts (offsets has nchunks + 1 entries): a damaged header,
    // field or run length must never walk the input cursor past it (the output side is guarded below)
    const uint64_t slen64 = a.offsets[chunk + 1] - off_c;
    const uint32_t stream_len = slen64 < 0xffffffffull ? (uint32_t)slen64 : 0xffffffffu;
    U* const o = (U*)a.out + chunk * (uint64_t)a.chunk_len;
    const uint64_t cs = a.col_stride;
    U* const cm0 = (U*)a.out + (cs ? chunk * (uint64_t)(a.chunk_len / (uint32_t)a.D) : 0);   // column 0 at this chunk's first row
    uint8_t* const lds = smem + (size_t)(threadIdx.x >> a.log2DP) * a.lds_group_stride;

    // ---- 8-byte stream header (format.h:48-62)
    uint32_t groups_left, remaining, pos;
    if (a.norle) {                                   // format.h:65-86; sprintz_delta.cpp:803-807, :832
        // norle == 2: compress8b_rowmajor_xff's 8-byte header, a u64 len whose bytes 6..7 hold ndims (sprintz_xff.cpp:58-63)
        const uint32_t len = load_u32_any(s);
        const uint32_t ndo = a.norle == 2 ? 6u : 4u;
        const uint32_t nd = load_u8(s + ndo) | (load_u8(s + ndo + 1) << 8);
        if ((int)nd != D || len > a.chunk_len) {
            if (lane_d == 0 && a.rets) a.rets[chunk] = kErrCorrupt;
            return;
        }
        groups_left = len < 128u ? 0u : len / (16u * (uint32_t)D);
        remaining = len - groups_left * 16u * (uint32_t)D;
        pos = a.norle == 2 ? 8u : 6u;
    } else if (!a.noheader) {
        const uint32_t w0 = load_u32_any(s), w1 = load_u32_any(s + 4);
        groups_left = w0;
        remaining = w1 & 0xffffu;
        pos = 8;
        if ((int)(w1 >> 16) != D) {
            if (lane_d == 0 && a.rets) a.rets[chunk] = kErrCorrupt;
            return;
        }
    } else {
        groups_left = a.nh_ngroups;
        remaining = a.nh_remaining;
        pos = 0;
    }

    const uint32_t hdr_bytes = (2u * (uint32_t)D * HB + 7u) >> 3;
    const uint32_t blk_elems = 8u * (uint32_t)D;
    // a damaged header must not make the loop spin: every group of a valid stream holds at
    // least one non-empty slot, except the one that closes the stream
    bool corrupt = groups_left > a.chunk_len / blk_elems + 2u || pos > stream_len;
    if (corrupt) groups_left = 0;

    // per-column predictor state (all start at 0: sprintz_xff_rle.cpp:149-152)
    uint32_t pv[CPL];
    int pd[CPL], ctr[CPL];
    uint32_t nb0[CPL], nb1[CPL];     // nbits of the two slots of the current group
    uint32_t tot0 = 0, tot1 = 0;
#pragma unroll
    for (int k = 0; k < CPL; k++) { pv[k] = 0; pd[k] = 0; ctr[k] = 0; nb0[k] = 0; nb1[k] = 0; }

    uint32_t out_elems = 0;
    int slot = 2;
    uint32_t run_left = 0;
    uint32_t qmax[CPL];
    uint64_t qsum[CPL];
#pragma unroll
    for (int k = 0; k < CPL; k++) { qmax[k] = 0; qsum[k] = 0; }

    for (;;) {
        uint32_t z[8][CPL];
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int k = 0; k < CPL; k++) z[i][k] = 0;

        bool have = false;
        bool run_block = false;                  // this block is one of a RUN
        if (run_left > 0) {                      // inside a RUN: zero errors (sprintz_xff_rle.cpp:828-958)
            run_left--;
            have = true;
            run_block = true;
        } else {
            for (;;) {
                if (slot == 2) {
                    if (groups_left == 0) break;
                    if (hdr_bytes > stream_len - pos) { corrupt = true; break; }
                    groups_left--;
                    // group header: 2*D fields of HB bits, LSB-first (sprintz_xff_rle.cpp:713-735)
                    uint32_t s0 = 0, s1 = 0;
#pragma unroll
                    for (int k = 0; k < CPL; k++) {
                        const int col = lane_d * CPL + k;
                        uint32_t f0 = 0, f1 = 0;
                        if (col < D) {
                            f0 = fetch_bits(s + pos, (uint32_t)col * HB, HB);
                            f1 = fetch_bits(s + pos, (uint32_t)(D + col) * HB, HB);
                        }
                        nb0[k] = f0 == (uint32_t)(W - 1) ? (uint32_t)W : f0;   // :747-749,763-765
                        nb1[k] = f1 == (uint32_t)(W - 1) ? (uint32_t)W : f1;
                        s0 += nb0[k];
                        s1 += nb1[k];
                    }
                    // both slot totals in one butterfly
                    const uint32_t both = group_sum(s0 | (s1 << 16), DP);
                    tot0 = both & 0xffffu;
                    tot1 = both >> 16;
                    pos += hdr_bytes;
                    slot = 0;
                }
                const uint32_t total = slot ? tot1 : tot0;
                if (total == 0 && a.norle) {     // a block of zeros: no payload at all
                    slot++;
                    have = true;
                    break;
                } else if (total == 0) {         // RUN slot: varint length in blocks (:829-833)
                    if (stream_len - pos < 2u) {     // room for the longest run length, or the stream is damaged
                        if (stream_len == pos || (load_u8(s + pos) & 0x80u)) { corrupt = true; break; }
                    }
                    const uint32_t b0 = load_u8(s + pos);
                    uint32_t len = b0 & 0x7fu;
                    if (b0 & 0x80u) { len |= load_u8(s + pos + 1) << 7; pos += 2; }
                    else pos += 1;
                    slot++;
                    if (len > 0) { run_left = len - 1; have = true; run_block = true; break; }
                    // len == 0: padding slot, look at the next one
                } else {                         // packed block
                    uint32_t cur[CPL], lane_bits = 0;
#pragma unroll
                    for (int k = 0; k < CPL; k++) { cur[k] = slot ? nb1[k] : nb0[k]; lane_bits += cur[k]; }
                    uint32_t tot_unused;
                    uint32_t off = group_excl_scan(lane_bits, lane_d, DP, tot_unused);
                    if ((LOWDIM ? total : ((total + 7u) >> 3) << 3) > stream_len - pos) { corrupt = true; break; }
                    if constexpr (!LOWDIM) {
                        // row r: LSB-first bit stream of the D fields, padded to a byte (:961-990)
                        const uint32_t row_bits = ((total + 7u) >> 3) << 3;
#pragma unroll
                        for (int k = 0; k < CPL; k++) {
#pragma unroll
                            for (int i = 0; i < 8; i++)
                                z[i][k] = fetch_bits(s + pos, (uint32_t)i * row_bits + off, cur[k]);
                            off += cur[k];
                        }
                        pos += row_bits;         // 8 rows * row_bytes
                    } else {
                        // column-major: nbits[d] bytes per column (sprintz_delta_lowdim.cpp:561-603)
#pragma unroll
                        for (int k = 0; k < CPL; k++) {
#pragma unroll
                            for (int i = 0; i < 8; i++)
                                z[i][k] = fetch_bits(s + pos, off * 8u + (uint32_t)i * cur[k], cur[k]);
                            off += cur[k];
                        }
                        pos += total;
                    }
                    slot++;
                    have = true;
                    break;
                }
            }
        }
        if (!have || corrupt) break;
        if (out_elems + blk_elems > a.chunk_len) { corrupt = true; break; }   // never write outside the chunk slot

        // ---- zigzag^-1 + forecast recurrence, lane-local down each column (:993-1150)
        uint32_t v[8][CPL];
#pragma unroll
        for (int k = 0; k < CPL; k++) {
            int coef = FIRE ? fire_coef<W, LOWDIM>(ctr[k]) : 0;
            if constexpr (FIRE && W == 16 && !LOWDIM) {
                if (a.quirk && run_block) coef = fire_coef_ref_run16(ctr[k], lane_d * CPL + k);
            }
            int grad = 0;
            uint32_t pvk = pv[k];
            int pdk = pd[k];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int err = unzigzag(z[i][k]);
                const int pred = FIRE ? fire_predict<W, LOWDIM>(pdk, coef) : 0;
                const int delta = sext<W>(err + pred);
                if (FIRE && (i & 1)) grad += sign_times(err, pdk);
                pvk = (pvk + (uint32_t)delta) & MASK;
                pdk = delta;
                v[i][k] = (!FIRE && a.raw) ? z[i][k] : pvk;     // raw: the packed bits ARE the samples (sprintz_delta.cpp:143-146)
            }
            pv[k] = pvk;
            pd[k] = pdk;
            if (FIRE) ctr[k] = wrap_counter<W>(ctr[k] + (sext<W>(grad) >> 2));   // :1120-1128
            if constexpr (Q != 0) {              // the query functor sees every decoded row (sprintz_xff_rle_query.hpp:346-596)
                uint32_t bs = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    qmax[k] = v[i][k] > qmax[k] ? v[i][k] : qmax[k];
                    bs += v[i][k];
                }
                qsum[k] += bs;
            }
        }

        // ---- store the 8 x D block (contiguous 8*D*ESZ bytes of the output)
        U* const ob = o + out_elems;
        if constexpr (Q == kQueryReduceOnly) {
            (void)ob;
        } else if (cs) {
            const uint32_t r0 = out_elems / (uint32_t)D;
#pragma unroll
            for (int k = 0; k < CPL; k++) {
                const int col = lane_d * CPL + k;
                if (col < D) {
                    U* const cp = cm0 + (uint64_t)col * cs + r0;
#pragma unroll
                    for (int i = 0; i < 8; i++) cp[i] = (U)v[i][k];
                }
            }
        } else if (LOWDIM && D == 1) {
            // one column: the block's 8 samples are contiguous -- one store, not eight
            if (lane_d == 0) {
                typedef uint32_t v2 __attribute__((ext_vector_type(2)));
                typedef uint32_t v4 __attribute__((ext_vector_type(4)));
                typedef v2 __attribute__((aligned(1), may_alias)) v2u;
                typedef v4 __attribute__((aligned(1), may_alias)) v4u;
                if constexpr (W == 8) {
                    v2 t;
                    t.x = v[0][0] | (v[1][0] << 8) | (v[2][0] << 16) | (v[3][0] << 24);
                    t.y = v[4][0] | (v[5][0] << 8) | (v[6][0] << 16) | (v[7][0] << 24);
                    *(v2u*)ob = t;
                } else {
                    v4 t;
                    t.x = v[0][0] | (v[1][0] << 16);
                    t.y = v[2][0] | (v[3][0] << 16);
                    t.z = v[4][0] | (v[5][0] << 16);
                    t.w = v[6][0] | (v[7][0] << 16);
                    *(v4u*)ob = t;
                }
            }
        } else if (a.vec_store) {
            U* const l = (U*)lds;
#pragma unroll
            for (int k = 0; k < CPL; k++) {
                const int col = lane_d * CPL + k;
                if (col < D) {
#pragma unroll
                    for (int i = 0; i < 8; i++) l[i * D + col] = (U)v[i][k];
                }
            }
            wave_lds_sync();
            const uint32_t nunits = (blk_elems * ESZ) >> 4;
            for (uint32_t u = (uint32_t)lane_d; u < nunits; u += (uint32_t)DP)
                ((uint4*)ob)[u] = ((const uint4*)lds)[u];
            wave_lds_sync();
        } else {
#pragma unroll
            for (int k = 0; k < CPL; k++) {
                const int col = lane_d * CPL + k;
                if (col < D) {
#pragma unroll
                    for (int i = 0; i < 8; i++) ob[i * D + col] = (U)v[i][k];
                }
            }
        }
        out_elems += blk_elems;
    }

    // ---- verbatim tail (:1171)
    if (!corrupt && (out_elems + remaining > a.chunk_len || (uint64_t)remaining * ESZ > (uint64_t)(stream_len - pos))) corrupt = true;
    if constexpr (Q != 0) {
        // the verbatim tail continues the row-major order: element e sits in column e % D
        // (out_elems is a multiple of 8*D)
        if (!corrupt) {
            const uint8_t* t = s + pos;
#pragma unroll
            for (int k = 0; k < CPL; k++) {
                const int col = lane_d * CPL + k;
                if (col >= D) continue;
                for (uint32_t e = (uint32_t)col; e < remaining; e += (uint32_t)D) {
                    const uint32_t x = ESZ == 1 ? load_u8(t + e) : (load_u8(t + 2 * e) | (load_u8(t + 2 * e + 1) << 8));
                    qmax[k] = x > qmax[k] ? x : qmax[k];
                    qsum[k] += x;
                }
                if (a.qres) a.qres[chunk * (uint64_t)D + (uint64_t)col] = a.qop == 1 ? (uint64_t)qmax[k] : qsum[k];
            }
        }
    }
    if (!corrupt && Q != kQueryReduceOnly && cs) {
        const uint8_t* t = s + pos;
        const uint32_t r0 = out_elems / (uint32_t)D;
        for (uint32_t e = (uint32_t)lane_d; e < remaining; e += (uint32_t)DP) {
            const uint32_t x = ESZ == 1 ? load_u8(t + e) : (load_u8(t + 2 * e) | (load_u8(t + 2 * e + 1) << 8));
            cm0[(uint64_t)(e % (uint32_t)D) * cs + r0 + e / (uint32_t)D] = (U)x;
        }
    } else if (!corrupt && Q != kQueryReduceOnly) {
        const uint8_t* t = s + pos;
        uint8_t* d = (uint8_t*)(o + out_elems);
        copy_verbatim(t, d, remaining * ESZ, (uint32_t)lane_d, (uint32_t)DP);
    }
    if (lane_d == 0 && a.rets) a.rets[chunk] = corrupt ? kErrCorrupt : (int64_t)out_elems + remaining;
}

}  // namespace sprintz

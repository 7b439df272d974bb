// decode_row.h -- the COLUMN-GROUP-SEQUENTIAL decoder of the DELTA codec, general (row-major) payload layout, for large batches
// (decompress_rowmajor_delta_rle, sprintz_delta_rle.cpp:418-772).  Same samples and return values as decode_kernel.h.
//
// Round 6 built two block-parallel forms first (decode_blk.h) and measured why they lose: the carry a column needs from the block above costs
// a pass of its own, and the serial phases (the header walk, the carry scan) hold a workgroup's other waves at barriers with 38 KB of LDS held.
// This kernel keeps what worked there -- a lane takes the 4 (2) fields of ONE output dword from one 32-bit window, undoes the zigzag inside the
// field fetch and keeps the running sum PACKED with an SDWA add per sample -- and drops what did not: a lane owns one dword-wide column group
// (4 uint8 / 2 uint16 columns) and walks its chunk's blocks IN ORDER, so
//   * the carry is the dword it stored last: free;
//   * the group walk is every lane's own: the U = row bytes / 4 lanes of a chunk parse their own columns' header fields (they need them
//     anyway), a wave scan of the width sums gives each its bit offset in the row and all of them the row length -- where the next block and
//     the next header start;
//   * there is no stream image in LDS and no barrier: a row's fields of mine are one aligned 8-byte load straight from global memory (L2 /
//     L1: a chunk's lanes read one contiguous stream front to back), all eight rows of a block in flight together.
// 64 / U chunks a wavefront (cfg3: 20 lanes a chunk, 3 chunks a wave); ~60 registers: the waves that hide the loads' latency fit.
// (A first form with 16 columns a lane -- 5 lanes a chunk -- had a quarter of the waves and 142 registers: 0.337 ms on cfg3, latency-bound.)
// Shapes (api.hip): delta codec, general layout, rows of whole dwords, U <= 64, 4-byte aligned container and output, any chunk length.
#pragma once

#include "decode_fast.h"
#include "decode_blk.h"

namespace sprintz {

struct RowDecGeom {
    uint32_t U;          // dwords per row = lanes per chunk
    uint32_t G;          // chunks per wavefront = 64 / U
    uint32_t invU;       // ceil(2^16 / U): lane / U == (lane * invU) >> 16 for lane < 64
    uint32_t ok;
};

inline RowDecGeom row_dec_geom(uint32_t esz, uint32_t chunk_len, uint32_t D)
{
    RowDecGeom g{};
    const uint32_t rowbytes = D * esz;
    if (rowbytes % 4u || ((uint64_t)chunk_len * esz) % 4u || chunk_len < 16u * D) return g;
    g.U = rowbytes / 4u;
    if (g.U > 64u) return g;
    g.G = 64u / g.U;
    g.invU = (65536u + g.U - 1u) / g.U;
    g.ok = 1u;
    return g;
}

// An 8-byte window at bit address `bit` of the byte stream that starts `off` bytes into the 4-byte aligned container `comp` (global memory):
// one aligned 8-byte load; v_alignbit by sh gives the 32 bits at `bit`.  Offsets are 32-bit (the launch checks that the container and the
// output are below 4 GB): the address is the kernel's uniform pointer + a 32-bit lane offset -- the loads stay GLOBAL loads of the
// scalar-base form, and a row's address is two vector instructions.
struct Win2 { uint32_t lo, hi; };
__device__ __forceinline__ Win2 gwin_load(const uint8_t* comp, uint32_t off, uint32_t bit)
{
    const uint32_t A = off + (bit >> 3);
    const uint32_t* q = (const uint32_t*)(comp + (A & ~3u));
    Win2 w;
    w.lo = q[0];
    w.hi = q[1];
    return w;
}
__device__ __forceinline__ uint32_t gwin_shift(uint32_t off, uint32_t bit) { return (((off + (bit >> 3)) & 3u) << 3) + (bit & 7u); }
__device__ __forceinline__ uint32_t gbits32(const uint8_t* comp, uint32_t off, uint32_t bit)
{
    const Win2 w = gwin_load(comp, off, bit);
    return __builtin_amdgcn_alignbit(w.hi, w.lo, gwin_shift(off, bit));
}

template <int W>
__global__ void __launch_bounds__(256) decode_row_kernel(DecodeArgs a, RowDecGeom g)
{
    constexpr int HB = Elem<W>::HB;
    constexpr int ESZ = W / 8;
    constexpr int FPD = 4 / ESZ;                 // fields (columns) per lane: one output dword a row
    constexpr uint32_t FM = (1u << HB) - 1u;

    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t cg = (lane * g.invU) >> 16, u = lane - cg * g.U;
    const uint64_t chunk = ((uint64_t)blockIdx.x * 4u + (tid >> 6)) * g.G + cg;
    const bool exists = cg < g.G && chunk < a.nchunks;
    const uint32_t D = (uint32_t)a.D, blk = 8u * D, rowbytes = D * ESZ;
    const uint32_t hdr_bytes = (2u * D * HB + 7u) >> 3;
    const uint32_t seg0 = cg * g.U;              // my chunk's first lane in the wavefront
    const uint32_t seg_last = seg0 + g.U - 1u < 63u ? seg0 + g.U - 1u : 63u;

    const uint64_t off64 = exists ? a.offsets[chunk] : 0ull;
    const uint64_t slen64 = exists ? a.offsets[chunk + 1] - off64 : 0ull;
    const uint32_t slen = slen64 < 0x0fffffffull ? (uint32_t)slen64 : 0x0fffffffu;
    const uint32_t off = (uint32_t)off64;        // (the launch checked: the whole container is below 4 GB)
    const uint8_t* const s = a.comp + off;
    uint8_t* const out8 = (uint8_t*)a.out;
    const uint32_t o = (uint32_t)chunk * a.chunk_len * ESZ + u * 4u;      // byte offset of my dword of the chunk's first row (the output is below 4 GB too)

    // ---- 8-byte stream header (format.h:48-62)
    // (offsets are the caller's data: a stream that does not lie inside the first 4 GB of the container is never touched)
    bool corrupt = !exists || slen < 8u || off64 + slen64 >= 0xfffffff0ull;
    uint32_t groups_left = 0, remaining = 0;
    if (!corrupt) {
        const uint32_t w0 = gbits32(a.comp, off, 0), w1 = gbits32(a.comp, off, 32);
        groups_left = w0;
        remaining = w1 & 0xffffu;
        // a damaged header must not make the loop spin: every group of a valid stream holds at least one non-empty slot, except the last
        corrupt = (w1 >> 16) != D || groups_left > a.chunk_len / blk + 2u;
    }
    if (corrupt) groups_left = 0;

    uint32_t pos = 8u, out_blocks = 0;
    uint32_t last = 0u;                          // the dword above: the chunk starts from zero (:61-63)
    const uint32_t max_blocks = a.chunk_len / blk;
    constexpr uint32_t HM = (1u << (FPD * HB)) - 1u;

    // A block's 8 windows.  The lanes of a chunk read a ROW together -- lane i the aligned dwords i and i + 1 of the row: one coalesced
    // request a row instead of 64 scattered 8-byte ones (the first form of this kernel was bound by the texture addresser: TA busy 87 % of
    // the launch) -- and every lane picks the pair its fields start in from the lane that holds it (ds_bpermute: my fields start Bp bits
    // into the row, never behind my own dword).  cb: the block's first byte in the stream; R: bytes a row.
    auto load_rows = [&](Win2 (&raw)[8], uint32_t& phases, uint32_t cb, uint32_t R) {
        uint32_t A = off + cb;
        phases = 0;
        // (a lane past the row's end reads the row's last dwords again -- no lane leaves the instruction, and nothing is read more than
        //  11 bytes past a row, i.e. past the stream + its 16 bytes of slack: the clamp is the block's, whatever a row's byte phase)
        const uint32_t cap4 = (R + 3u) & ~3u, mine4 = 4u * u < cap4 ? 4u * u : cap4;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t* q = (const uint32_t*)(a.comp + ((A & ~3u) + mine4));
            raw[r].lo = q[0];
            raw[r].hi = q[1];
            phases |= (A & 3u) << (2 * r);                               // the rows' byte phases, two bits each
            A += R;
        }
    };
    const bool quads = (g.U & 3u) == 0u;         // rows of whole 16-byte pieces: a quad of lanes transposes 4 rows x 4 dwords and stores 16 bytes a lane
    const bool odd1 = (lane & 1u) != 0u, odd2 = (lane & 2u) != 0u;
    auto decode_rows = [&](const Win2 (&raw)[8], uint32_t phases, uint32_t Bp, const uint32_t (&nb4)[FPD], uint32_t ob) {
        Win2 wv[8];
        uint32_t X[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {                                    // bit offset of my fields in the row's aligned dwords; the pair they start in, from its lane
            X[r] = (((phases >> (2 * r)) & 3u) << 3) + Bp;
            const int src = (int)((seg0 + (X[r] >> 5)) << 2);
            wv[r].lo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)raw[r].lo);
            wv[r].hi = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)raw[r].hi);
        }
        uint32_t fs[FPD], fs1[FPD], wm[FPD], w1[FPD];
        uint32_t acc = 0;
#pragma unroll
        for (int f = 0; f < FPD; f++) {
            fs[f] = acc;
            fs1[f] = acc + 1u;
            w1[f] = nb4[f] != 0u ? 1u : 0u;
            wm[f] = nb4[f] - w1[f];
            acc += nb4[f];
        }
        uint32_t rows[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t t = __builtin_amdgcn_alignbit(wv[r].hi, wv[r].lo, X[r]);     // (the shift's low 5 bits count)
            uint32_t row = 0;
#pragma unroll
            for (int f = 0; f < FPD; f++) {
                const uint32_t e = __builtin_amdgcn_ubfe(t, fs1[f], wm[f]) ^ (uint32_t)__builtin_amdgcn_sbfe((int)t, fs[f], w1[f]);
                if (f == 0) field_acc<W, 0>(row, last, e);
                else if (f == 1) field_acc<W, 1>(row, last, e);
                else if (f == 2) field_acc<W, (FPD > 2 ? 2 : 1)>(row, last, e);
                else field_acc<W, (FPD > 2 ? 3 : 1)>(row, last, e);
            }
            last = row;
            rows[r] = row;
        }
        if (quads) {
            typedef uint32_t v4 __attribute__((ext_vector_type(4)));
            const uint32_t q16 = ob - 4u * (u & 3u);                     // the quad's 16-byte piece of row 0
#pragma unroll
            for (int h = 0; h < 2; h++) {                                // rows 4 h .. 4 h + 3: (lane j, row k) -> (lane k, dword j), two DPP butterfly stages
                uint32_t v[4] = {rows[4 * h], rows[4 * h + 1], rows[4 * h + 2], rows[4 * h + 3]};
#pragma unroll
                for (int k = 0; k < 4; k += 2) {
                    const uint32_t send = odd1 ? v[k] : v[k + 1];
                    const uint32_t recv = (uint32_t)__builtin_amdgcn_mov_dpp((int)send, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
                    if (odd1) v[k] = recv; else v[k + 1] = recv;
                }
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const uint32_t send = odd2 ? v[k] : v[k + 2];
                    const uint32_t recv = (uint32_t)__builtin_amdgcn_mov_dpp((int)send, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
                    if (odd2) v[k] = recv; else v[k + 2] = recv;
                }
                // lane j of the quad now holds row 4 h + j: dwords of the quad's lanes 0 .. 3
                __builtin_nontemporal_store(v4{v[0], v[1], v[2], v[3]}, (v4*)(out8 + (q16 + (uint32_t)(4 * h + (int)(u & 3u)) * rowbytes)));
            }
        } else {
#pragma unroll
            for (int r = 0; r < 8; r++) __builtin_nontemporal_store(rows[r], (uint32_t*)(out8 + (ob + (uint32_t)r * rowbytes)));
        }
    };

    // the NEXT group's header fields, requested before this group's stores go out: on this hardware a wait for a load is a wait for every
    // store issued before it (one counter), so whatever a group needs is asked for in front of the group before's stores
    Win2 pre0 = {0u, 0u}, pre1 = {0u, 0u};
    uint32_t pre_pos = 0xffffffffu;

    while (__ballot(groups_left != 0u) != 0ull) {
        const bool act = groups_left != 0u;
        const uint32_t hpos = act ? pos : 0u;
        bool bad = act && hdr_bytes > slen - hpos;
        // ---- my columns' fields of both slots: FPD * HB bits each at header bit u * FPD * HB (+ D * HB)
        const uint32_t hb0 = hpos * 8u + u * (uint32_t)(FPD * HB), hb1 = hb0 + D * HB;
        Win2 h0 = pre0, h1 = pre1;
        if (act && !bad && pre_pos != hpos) {
            h0 = gwin_load(a.comp, off, hb0);
            h1 = gwin_load(a.comp, off, hb1);
        }
        uint32_t hf[2] = {0u, 0u};
        if (act && !bad) {
            hf[0] = __builtin_amdgcn_alignbit(h0.hi, h0.lo, gwin_shift(off, hb0)) & HM;
            hf[1] = __builtin_amdgcn_alignbit(h1.hi, h1.lo, gwin_shift(off, hb1)) & HM;
        }
        pre_pos = 0xffffffffu;
        uint32_t nbs[2][FPD], mine = 0;
#pragma unroll
        for (int sl = 0; sl < 2; sl++) {
            uint32_t t = 0;
#pragma unroll
            for (int f = 0; f < FPD; f++) {
                const uint32_t fv = (hf[sl] >> (HB * f)) & FM;
                nbs[sl][f] = fv + ((fv + 1u) >> HB);                      // W - 1 means W (:747-749)
                t += nbs[sl][f];
            }
            mine |= t << (16 * sl);
        }
        // ---- the wavefront scans the width sums (both slots in one word: a row is <= 64 x 32 bits); a chunk's lanes take their segment's part
        uint32_t wave_total;
        const uint32_t excl = group_scan<64>(mine, (int)lane, wave_total);
        const uint32_t incl = excl + mine;
        const uint32_t base = (uint32_t)__shfl((int)excl, (int)seg0, 64);         // the sums in front of my chunk's first lane
        const uint32_t total = (uint32_t)__shfl((int)incl, (int)seg_last, 64) - base;
        const uint32_t before = excl - base;
        const uint32_t S0 = total & 0xffffu, S1 = total >> 16;
        const uint32_t pay0 = ((S0 + 7u) >> 3) << 3, pay1 = ((S1 + 7u) >> 3) << 3;

        // ---- the common group: two packed blocks that fit the stream and the chunk -- sixteen windows and the next header in flight, then the rows
        const bool fastg = act && !bad && S0 != 0u && S1 != 0u && hdr_bytes + pay0 + pay1 <= slen - hpos && out_blocks + 2u <= max_blocks;
        if (__ballot(fastg) != 0ull) {
            if (fastg) {
                const uint32_t c0 = hpos + hdr_bytes, c1 = c0 + pay0, npos = c1 + pay1;
                const uint32_t B0 = before & 0xffffu, B1 = before >> 16;
                Win2 wa[8], wb[8];
                uint32_t pa, pb;
                load_rows(wa, pa, c0, pay0 >> 3);
                load_rows(wb, pb, c1, pay1 >> 3);
                if (groups_left > 1u && hdr_bytes <= slen - npos) {
                    pre0 = gwin_load(a.comp, off, npos * 8u + u * (uint32_t)(FPD * HB));
                    pre1 = gwin_load(a.comp, off, npos * 8u + u * (uint32_t)(FPD * HB) + D * HB);
                    pre_pos = npos;
                }
                const uint32_t ob = o + out_blocks * blk * ESZ;
                decode_rows(wa, pa, B0, nbs[0], ob);
                decode_rows(wb, pb, B1, nbs[1], ob + blk * ESZ);
                out_blocks += 2u;
                pos = npos;
                groups_left -= 1u;
            }
        }
        // ---- a run, a padding slot, the chunk's end, damage: slot by slot
        if (__ballot(act && !fastg) != 0ull) {
            if (act && !fastg) {
                uint32_t cur = hpos + hdr_bytes;
#pragma unroll
                for (int sl = 0; sl < 2; sl++) {
                    const uint32_t S = sl ? S1 : S0;
                    const uint32_t Bp = sl ? before >> 16 : before & 0xffffu;
                    if (bad) continue;
                    if (S == 0u) {                                           // RUN slot: length in blocks, 1 or 2 bytes (:829-833)
                        if (slen - cur < 2u) bad = slen == cur || (s[cur] & 0x80u) != 0u;
                        if (!bad) {
                            const uint32_t b0 = s[cur];
                            uint32_t len = b0 & 0x7fu;
                            if (b0 & 0x80u) { len |= (uint32_t)s[cur + 1u] << 7; cur += 2u; }
                            else cur += 1u;
                            if (out_blocks + len > max_blocks) bad = true;
                            else {                                           // the dword above, 8 times a block (zero deltas)
                                for (uint32_t b = 0; b < len; b++) {
                                    const uint32_t ob = o + (out_blocks + b) * blk * ESZ;
#pragma unroll
                                    for (int r = 0; r < 8; r++) __builtin_nontemporal_store(last, (uint32_t*)(out8 + (ob + (uint32_t)r * rowbytes)));
                                }
                                out_blocks += len;
                            }
                        }
                    } else {                                                 // a packed block: 8 rows of ceil(S / 8) bytes; my fields start Bp bits into each row
                        const uint32_t pay = sl ? pay1 : pay0;
                        if (pay > slen - cur || out_blocks >= max_blocks) bad = true;
                        else {
                            Win2 wv[8];
                            uint32_t pv;
                            load_rows(wv, pv, cur, pay >> 3);
                            if (sl == 0) decode_rows(wv, pv, Bp, nbs[0], o + out_blocks * blk * ESZ);
                            else decode_rows(wv, pv, Bp, nbs[1], o + out_blocks * blk * ESZ);
                            out_blocks += 1u;
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

    // ---- the verbatim tail (:1171): my share of its dwords, the last odd bytes by the chunk's first lane
    if (exists) {
        if (!corrupt && (out_blocks * blk + remaining > a.chunk_len || (uint64_t)remaining * ESZ > (uint64_t)(slen - pos))) corrupt = true;
        if (!corrupt) {
            const uint32_t tb = remaining * ESZ;
            uint8_t* const d = out8 + ((uint32_t)chunk * a.chunk_len + out_blocks * blk) * ESZ;
            for (uint32_t i = u; i < tb >> 2; i += g.U) *(uint32_t*)(d + 4u * i) = gbits32(a.comp, off + pos, 32u * i);
            if (u == 0) for (uint32_t i = tb & ~3u; i < tb; i++) d[i] = s[pos + i];
        }
        if (u == 0 && a.rets) a.rets[chunk] = corrupt ? kErrCorrupt : (int64_t)out_blocks * blk + remaining;
    }
}

hipError_t launch_decode_row(int w, unsigned grid, hipStream_t st, const DecodeArgs& a, const RowDecGeom& g);

}  // namespace sprintz

// decode_lat.h -- the LATENCY decoder: one WORKGROUP (256 lanes) per chunk, for batches too small to fill the chip with
// decode_fast.h's one-lane-per-column mapping (a single drop-in call; one GPU's 1 250-chunk share of BASELINE config 4's
// 10 000-chunk batches on eight).  Same streams, same samples, same return values as decode_fast.h / decode_kernel.h
// (sprintz_xff_rle.cpp:569-1179, sprintz_delta_rle.cpp:418-772; the low-dim layouts of sprintz_{delta,xff}_lowdim.cpp as template
// parameter LOW), headered RLE streams, ndims <= 64, chunks of at most 16 KB in batches (several workgroups a CU) and up to what a
// workgroup's 150 KB of LDS holds (~40 KB of uint16, ~24 KB of uint8) for single calls and batches of at most 64 chunks (api.hip: lat_chunk_fits).
//
// decode_fast.h walks a chunk's 40 groups in 40 dependent steps of ~600 wave-instructions each: 50 us a chunk however few
// chunks there are (a lone wave issues an instruction every 4 .. 8 cycles).  Only two things in the format are serial:
//   (1) WHERE a group starts -- the sum of the header fields of all groups before it;
//   (2) the forecast recurrence down a column -- delta[i] = err[i] + ((delta[i-1] * coef) >> W), coef changing per block.
// Everything else (field offsets, bit extraction, zigzag^-1, the running sum of deltas, the transpose to row-major, the
// stores) is independent per block or a prefix sum.  So, per chunk, with the whole stream parked in LDS, the four waves of
// the workgroup form a pipeline (hand-offs through LDS words, no workgroup barrier inside it):
//   A  wave 0 walks the group headers: field sum by DPP, run lengths, cursor -> grp[g] = (stream position, first output
//      block, end block); publishes the number of groups walked;
//   B  waves 1..2 (1..3 without FIRE), a DP-lane group per stream group, as soon as A has published it: header scan, bit
//      fields -> E = err << W (err itself for 8-bit delta coding) into err[block][column][8 rows]; RUN slots -> zeros (a run
//      block IS a block of zero errors: the recurrence below then does what sprintz_xff_rle.cpp:828-958 replays, and its
//      gradient is 0 so the counters stand still); publishes its rounds;
//   C  FIRE only, wave 3: lane d < D runs column d's recurrence over the blocks whose errors are there, in place;
// then, behind one barrier,
//   D  256 lanes: sums of every block's deltas -> prefix over the blocks of a column (mod 2^W) -> samples to a row-major
//      image in LDS (over the dead stream) -> whole 16-byte stores.
// A and C are chains of ~250 and ~130 ns a group / a block on MI355X: side by side they take ~11 us for uint16 x 8 columns x
// 640 rows instead of the 50 of the lane-per-column walk (measured: tools/lat_phases.py).
// It costs about twice the instructions per chunk of decode_fast.h, so it is for batches that leave the chip empty anyway.
#pragma once

#ifndef SPRINTZ_LAT_ROTATE
#define SPRINTZ_LAT_ROTATE 0           // round 5: roles rotated by workgroup number measured no different (35.5 vs 35.4 us at 1 250 chunks): off
#endif
#ifndef SPRINTZ_LAT_POLL_SLEEP
#define SPRINTZ_LAT_POLL_SLEEP 1          // s_sleep units (64 clocks) between two polls of a hand-off word
#endif

#include "decode_fast.h"

namespace sprintz {

constexpr uint32_t kLatMaxChunkBytes = 16u << 10;     // the stream, the error image and the tables of ONE chunk must fit LDS
constexpr uint32_t lat_align16(uint32_t x) { return (x + 15u) & ~15u; }
// LDS carve: [stream: strm_cap + 32 | grp: NB + 3 pairs of words | err: one int per block element | sum: one word per (column, block)]
struct LatCarve {
    uint32_t strm_cap, o_grp, o_err, o_sum, total;
};
inline LatCarve lat_carve(uint32_t bound_bytes, uint32_t chunk_len, uint32_t D)
{
    LatCarve c;
    const uint32_t nb = chunk_len / (8u * D);
    c.strm_cap = lat_align16(bound_bytes + 32u);
    c.o_grp = c.strm_cap + 32u;
    c.o_err = c.o_grp + lat_align16((nb + 3u) * 8u + 16u);
    c.o_sum = c.o_err + (nb + 1u) * 8u * D * 4u + 16u;      // (+1 block: phase C reads one block ahead)
    c.total = c.o_sum + lat_align16(D * (nb | 1u) * 4u + 16u);
    return c;
}

// LOW: the low-dim layout (D <= 4 at 8 bits, <= 2 at 16: sprintz_{delta,xff}_lowdim.cpp) -- column-major payload (a column's 8 values in
// nbits bytes), no row padding, untruncated FIRE coefficient (counter >> 1; 32-bit multiply at 16 bits).  Its errors travel
// unscaled (E = err): the general layout's "delta is the high half of prev_delta*coef + E" needs a 16-bit coefficient.
template <int W, bool FIRE, int DP, bool LOW = false>
__global__ void __launch_bounds__(256) decode_lat_kernel(DecodeArgs a, LatCarve cv)
{
    using U = typename Elem<W>::U;
    typedef int v4i __attribute__((ext_vector_type(4)));
    constexpr int HB = Elem<W>::HB;
    constexpr int ESZ = W / 8;
    constexpr int LOG2DP = DP == 4 ? 2 : DP == 8 ? 3 : DP == 16 ? 4 : DP == 32 ? 5 : 6;
    constexpr int T = 256 / DP;                            // DP-lane groups of a workgroup = lanes per column in the scan
    constexpr int LOG2T = 8 - LOG2DP;
    constexpr uint32_t G = 64 / DP;                        // stream groups a wave of phase B takes per round
    constexpr uint32_t NBW = FIRE ? 2u : 3u;               // waves of phase B
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ uint32_t info[8];                           // corrupt, groups, blocks, tail position, tail elements
#ifdef SPRINTZ_LAT_TIMING
    __shared__ uint64_t dbg[8];
#define LAT_DBG(k) do { if ((tid & 63u) == 0) dbg[k] = wall_clock64(); } while (0)
#else
#define LAT_DBG(k)
#endif
    __shared__ uint32_t ctl[8];                            // 0: groups A has published, 1: A is done, 2..4: rounds wave 1 + k of B has finished

    const uint32_t tid = threadIdx.x;
    const uint64_t chunk = blockIdx.x;
    const int D = a.D;
#ifdef SPRINTZ_LAT_TIMING                                  // experiment builds: phase durations (20 ns ticks, 12 bits each) instead of the return value
    uint64_t stamp[6];
    int nstamp = 0;
#define LAT_STAMP() stamp[nstamp++] = wall_clock64()
#else
#define LAT_STAMP()
#endif
    LAT_STAMP();
    const uint32_t hdr_bytes = (2u * (uint32_t)D * HB + 7u) >> 3;
    const uint32_t blk_elems = 8u * (uint32_t)D;
    const uint32_t NB = a.chunk_len / blk_elems;           // blocks a chunk's output can hold
    uint8_t* const strm = smem;
    uint2* const grp = (uint2*)(smem + cv.o_grp);
    int* const err = (int*)(smem + cv.o_err);
    uint32_t* const bsum = (uint32_t*)(smem + cv.o_sum);
    const uint32_t sbase = lds_addr(strm);

    // ---- 0: the stream, 16 bytes a lane, into LDS (from the 16-byte boundary below its first byte; the <= 15 bytes read past
    // its end are inside SPRINTZ_MI355X_READ_SLACK)
    const uint64_t off_c = a.offsets ? a.offsets[chunk] : a.one_off0;
    const uint64_t slen64 = (a.offsets ? a.offsets[chunk + 1] : a.one_off1) - off_c;
    const uint32_t shift = (uint32_t)((uintptr_t)(a.comp + off_c) & 15u);
    const uint32_t slen = slen64 < (uint64_t)(cv.strm_cap - 16u) ? (uint32_t)slen64 : cv.strm_cap - 16u;   // (a valid stream is shorter: strm_cap covers the bound)
    const uint32_t send = shift + slen;                    // LDS offset of the stream's end
    {
        const uint4* g = (const uint4*)(a.comp + (off_c - shift));
        const uint32_t n16 = (send + 15u) >> 4;
        for (uint32_t i = tid; i < n16; i += 256u) ((uint4*)strm)[i] = g[i];
        if (tid < 2u) ((uint4*)strm)[n16 + tid] = make_uint4(0, 0, 0, 0);     // the unaligned 32-bit windows read up to 7 bytes on
        if (tid < 8u) ctl[tid] = 0;
    }
    __syncthreads();
    LAT_STAMP();

    // the wave's ROLE (A: header walk, B: bit fields, C: recurrence), rotated by the workgroup's number: a workgroup's waves land on
    // the CU's four SIMDs in order, and with the roles fixed every chunk's recurrence wave -- the longest chain, 33 instructions a
    // block on 8 of 64 lanes -- sat on the same SIMD of its CU (round 5: 1 250 chunks = 5 workgroups a CU)
    const uint32_t wave = ((tid >> 6) + (SPRINTZ_LAT_ROTATE ? blockIdx.x : 0u)) & 3u;
    const int lane_d = (int)(tid & (uint32_t)(DP - 1));
    const bool col_ok = lane_d < D;
    const int colk = col_ok ? lane_d : D - 1;              // a lane past the last column stands in for it (uniform code)
    const uint32_t hbit0 = (uint32_t)colk * HB, hbit1 = (uint32_t)(D + colk) * HB;
    auto fields = [&](uint32_t r) -> uint32_t {            // both slots' nbits of this lane's column: slot 0 | slot 1 << 16
        const uint32_t hw0 = lds_rd32(r + (hbit0 >> 3)), hw1 = lds_rd32(r + (hbit1 >> 3));
        uint32_t f0 = __builtin_amdgcn_ubfe(hw0, hbit0 & 7u, HB), f1 = __builtin_amdgcn_ubfe(hw1, hbit1 & 7u, HB);
        f0 += (f0 == (uint32_t)(W - 1));                   // W-1 means W (:747-749)
        f1 += (f1 == (uint32_t)(W - 1));
        return col_ok ? (f0 | (f1 << 16)) : 0u;
    };
    auto run_length = [&](uint32_t at, uint32_t& nbytes) -> uint32_t {   // varint in blocks (:829-833)
        const uint32_t b0 = lds_rd8(at);
        uint32_t len = b0 & 0x7fu;
        nbytes = 1;
        if (b0 & 0x80u) { len |= lds_rd8(at + 1) << 7; nbytes = 2; }
        return len;
    };
    // (the hand-off words: DS operations of a wave execute in issue order, so a wave that sees a counter sees what was written
    //  before it; the fences keep the compiler from moving accesses across)
    // (plain LDS accesses: a `volatile __shared__` word compiles to system-scope FLAT loads -- 0.3 us a poll, measured)
    auto slot_bytes = [&](uint32_t tot) -> uint32_t { return LOW ? tot : ((tot + 7u) >> 3) * 8u; };   // payload bytes of a packed block
    auto publish = [&](uint32_t word, uint32_t value) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if ((tid & 63u) == 0) __hip_atomic_store(&ctl[word], value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto peek = [&](uint32_t word) -> uint32_t {           // wave-uniform by construction: into an SGPR, so that the polls branch on scalars
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&ctl[word], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    };

    if (wave == 0) {
        // ---- A: where every group starts and which output blocks it fills (the wave's DP-lane groups all compute the same)
        uint32_t pos = shift;
        const uint32_t w0 = lds_rd32(sbase + pos), w1 = lds_rd32(sbase + pos + 4u);
        const uint32_t groups = w0, remaining = w1 & 0xffffu;
        pos += 8u;
        bool corrupt = (int)(w1 >> 16) != D || slen < 8u;
        // a damaged header must not make the loop spin: every group takes at least its header and two slot bytes out of the stream
        if ((uint64_t)groups * (hdr_bytes + 2u) > slen64 || groups > NB + 2u) corrupt = true;
        uint32_t ob = 0, g = 0;
        if (!corrupt) {
            // (the walk is a chain of ~45 instructions a group: the common case -- both slots packed, everything in range -- runs
            //  straight through; RUN slots and every bound that fails share ONE branch)
            for (; g < groups; g++) {
                uint32_t tot_both;
                (void)group_scan<DP>(fields(sbase + (pos + hdr_bytes <= send ? pos : shift)), lane_d, tot_both);
                tot_both = (uint32_t)__builtin_amdgcn_readfirstlane((int)tot_both);
                const uint32_t tot0 = tot_both & 0xffffu, tot1 = tot_both >> 16;
                uint32_t npos = pos + hdr_bytes + slot_bytes(tot0) + slot_bytes(tot1), nob = ob + 2u;
                if (__builtin_expect(tot0 == 0 || tot1 == 0 || npos > send || nob > NB, 0)) {
                    if (pos + hdr_bytes > send) { corrupt = true; break; }
                    const uint32_t at0 = pos + hdr_bytes;
                    uint32_t len0 = 1, len1 = 1, bytes0 = slot_bytes(tot0), bytes1 = slot_bytes(tot1);
                    if (tot0 == 0) len0 = run_length(sbase + (at0 < send ? at0 : send), bytes0);
                    const uint32_t at1 = at0 + bytes0;
                    if (tot1 == 0) len1 = run_length(sbase + (at1 < send ? at1 : send), bytes1);
                    npos = at1 + bytes1;
                    nob = ob + len0 + len1;
                    if (npos > send || nob > NB) { corrupt = true; break; }   // cursor past the stream / more blocks than the chunk holds
                }
                // every lane writes the same words to the same place (no exec-mask round trip for "lane 0 only")
                grp[g] = make_uint2(pos | (ob << 16), nob);
                // release: the group's words are ordered before the counter that publishes them (the DS queue keeps one wave's
                // stores in order anyway; this keeps the COMPILER from sinking the grp store below the counter store)
                __hip_atomic_store(&ctl[0], g + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                ob = nob;
                pos = npos;
            }
        }
        const uint32_t out_left = a.chunk_len - ob * blk_elems;
        if (!corrupt && (remaining > out_left || pos + remaining * ESZ > send)) corrupt = true;
        if ((tid & 63u) == 0) { info[0] = corrupt ? 1u : 0u; info[1] = g; info[2] = ob; info[3] = pos; info[4] = remaining; }
        publish(1, 1u);
        LAT_DBG(0);
    } else if (wave <= NBW) {
        // ---- B: bit fields -> errors, a DP-lane group per stream group, in rounds of G groups a wave
        const uint32_t w = wave - 1u;
        for (uint32_t r = 0;; r++) {
            const uint32_t g0 = (r * NBW + w) * G;
            uint32_t ad;
            for (;;) {
                const uint32_t fin = peek(1);              // (read before the count: "done" then means the count is final)
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                ad = peek(0);
                if (fin || ad >= g0 + G) break;
                __builtin_amdgcn_s_sleep(SPRINTZ_LAT_POLL_SLEEP);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (g0 >= ad) break;                           // (A is done and left nothing for this wave)
            const uint32_t g = g0 + ((tid & 63u) >> LOG2DP);
            if (g < ad) {
                const uint32_t gw = grp[g].x;
                const uint32_t pos = gw & 0xffffu;
                uint32_t ob = gw >> 16;
                uint32_t tot_both;
                const uint32_t nb_both = fields(sbase + pos);
                const uint32_t excl = group_scan<DP>(nb_both, lane_d, tot_both);
                uint32_t at = pos + hdr_bytes;
#pragma unroll
                for (int slot = 0; slot < 2; slot++) {
                    const uint32_t tot = slot ? tot_both >> 16 : tot_both & 0xffffu;
                    const uint32_t off = slot ? excl >> 16 : excl & 0xffffu;
                    const uint32_t nb = slot ? nb_both >> 16 : nb_both & 0xffffu;
                    if (tot != 0) {
                        const uint32_t rb = (tot + 7u) >> 3;
                        if (col_ok) {
                            const uint32_t w1 = nb != 0 ? 1u : 0u, wm = nb - w1;      // widths of the sign bit and of the magnitude
                            int e[8];
                            if constexpr (LOW) {           // column-major: this column's 8 values in nb bytes behind the columns before it (:561-603)
                                const uint32_t base = sbase + at + off;                 // (off: a sum of byte counts here)
#pragma unroll
                                for (int i = 0; i < 8; i++) {
                                    const uint32_t bit = (uint32_t)i * nb;
                                    const uint32_t wd = lds_rd32(base + (bit >> 3));
                                    const uint32_t mag = __builtin_amdgcn_ubfe(wd, (bit & 7u) + 1u, wm);
                                    const int sgn = __builtin_amdgcn_sbfe((int)wd, bit & 7u, w1);
                                    e[i] = (int)(mag ^ (uint32_t)sgn);
                                }
                            } else {
                                uint32_t p = sbase + at + (off >> 3);
                                const uint32_t sh = off & 7u;
#pragma unroll
                                for (int i = 0; i < 8; i++) {
                                    const uint32_t wd = lds_rd32(p);
                                    const uint32_t mag = __builtin_amdgcn_ubfe(wd, sh + 1u, wm);      // zigzag^-1 = (z >> 1) ^ -(z & 1), straight from the window
                                    const int sgn = __builtin_amdgcn_sbfe((int)wd, sh, w1);
                                    const int x = (int)(mag ^ (uint32_t)sgn);
                                    e[i] = W == 16 ? (int)((uint32_t)x << 16) : (FIRE ? (int)((uint32_t)x << 8) : x);
                                    p += rb;
                                }
                            }
                            v4i* const q = (v4i*)(err + ((size_t)ob * (uint32_t)D + (uint32_t)lane_d) * 8u);
                            q[0] = v4i{e[0], e[1], e[2], e[3]};
                            q[1] = v4i{e[4], e[5], e[6], e[7]};
                        }
                        ob += 1u;
                        at += LOW ? tot : rb * 8u;
                    } else {
                        uint32_t nbytes;
                        const uint32_t len = run_length(sbase + at, nbytes);
                        if (col_ok) {
                            for (uint32_t j = 0; j < len; j++) {
                                v4i* const q = (v4i*)(err + ((size_t)(ob + j) * (uint32_t)D + (uint32_t)lane_d) * 8u);
                                q[0] = v4i{0, 0, 0, 0};
                                q[1] = v4i{0, 0, 0, 0};
                            }
                        }
                        ob += len;
                        at += nbytes;
                    }
                }
            }
            publish(2u + w, r + 1u);
            if (r == 0) LAT_DBG(wave);
        }
        LAT_DBG(2 + wave);
    } else if constexpr (FIRE) {
        // ---- C: the forecast recurrence, lane d = column d, over the blocks whose errors have arrived, in place
        // (E -> X with delta = X >> 16 at 16 bits; E -> delta at 8)
        // (a lane past the last column runs column 0 along: the same reads, the same arithmetic, the same bytes written to the same
        //  place -- no predicate in the loop)
        v4i* e = (v4i*)(err + (size_t)((tid & 63u) < (uint32_t)D ? (tid & 63u) : 0u) * 8u);
        const uint32_t bstride = (uint32_t)D * 2u;         // v4i per block
        int pd = 0, ctr = 0;
        uint32_t b = 0, gr = 0;
        for (;;) {
            // which blocks are there now?  (polled only when the lane has caught up: then a whole round of B at a time)
            uint32_t ready = b;
            bool finished = false;
            for (;;) {
                const uint32_t fin = peek(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const uint32_t ad = peek(0);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                // group gr is published: has its wave of B finished the round it falls in?  (whole rounds at a time)
                while (gr < ad) {
                    const uint32_t idx = gr / G, wv = idx % NBW, rr = idx / NBW;
                    if (peek(2u + wv) <= rr) break;
                    const uint32_t last = (idx + 1u) * G < ad ? (idx + 1u) * G : ad;      // the groups of that round that exist
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    ready = (uint32_t)__builtin_amdgcn_readfirstlane((int)grp[last - 1u].y);
                    gr = last;
                }
                if (ready > b) break;
                if (fin && gr >= ad) { finished = true; break; }
                __builtin_amdgcn_s_sleep(SPRINTZ_LAT_POLL_SLEEP);
            }
            if (finished) break;
            if (b == 0) LAT_DBG(6);
            // the tight part: blocks b .. ready-1, the next block's errors requested before this one's are used (the read past the
            // last ready block lands on bytes nobody waits for, inside the carve)
            auto one_block = [&](const v4i& c0, const v4i& c1) {
                const int E[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
                int X[8];
                const int coef = fire_coef<W, LOW>(ctr);
                int grad = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    if constexpr (LOW) {                   // plain errors, the low-dim predictor (sprintz_xff_lowdim.cpp:170-183)
                        if (i & 1) grad = mad24(sign_of(E[i]), pd, grad);
                        pd = sext<W>(E[i] + fire_predict<W, true>(pd, coef));
                    } else if constexpr (W == 16) {        // X = prev_delta*coef + E; delta = hi16(X): pd carries X (decode_fast.h)
                        if (i & 1) grad = mad_i16_hi(pd, sign_of(E[i]), grad);
                        pd = mad_i16_hi(pd, coef, E[i]);
                    } else {
                        if (i & 1) grad = mad24(sign_of(E[i]), pd, grad);
                        pd = __builtin_amdgcn_sbfe(mad24(pd, coef, E[i]), W, W);
                    }
                    X[i] = pd;
                }
                ctr = wrap_counter<W>(ctr + __builtin_amdgcn_sbfe(grad, 2, W - 2));       // sext_W(grad) >> 2 (:273-275)
                e[0] = v4i{X[0], X[1], X[2], X[3]};
                e[1] = v4i{X[4], X[5], X[6], X[7]};
                e += bstride;
            };
            // (two blocks a trip, each with its own registers: rotating one set costs five moves a block on a chain of 33 instructions)
            uint32_t n = ready - b;
            v4i a0 = e[0], a1 = e[1];
            while (n >= 2u) {
                const v4i b0 = e[bstride], b1 = e[bstride + 1];
                one_block(a0, a1);
                a0 = e[bstride];                           // (e has moved on: the block after the next one)
                a1 = e[bstride + 1];
                one_block(b0, b1);
                n -= 2u;
            }
            if (n) one_block(a0, a1);
            b = ready;
        }
        LAT_DBG(7);
    }
    __syncthreads();
    LAT_STAMP();
    const uint32_t nblk = info[2], tail_pos = info[3], remaining = info[4];
    // the call's last word: the return value, and for a single call on mapped host memory the caller's ticket behind every
    // lane's stores (the host polls that word instead of asking the runtime)
    auto finish = [&](int64_t r) {
        if (a.host_flag) { __threadfence_system(); __syncthreads(); }
        if (tid == 0) {
            if (a.rets) a.rets[chunk] = r;
            if (a.host_flag) {
                __threadfence_system();
                __hip_atomic_store(a.host_flag, a.host_ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    };
    if (info[0]) {                                         // nothing of a damaged stream is written
        finish(kErrCorrupt);
        return;
    }
    uint8_t* const obase = (uint8_t*)a.out + chunk * (uint64_t)a.chunk_len * ESZ;
    const uint32_t body_bytes = nblk * blk_elems * ESZ;
    // the verbatim tail (:1171) leaves now: phase D parks the samples where the stream lies
    for (uint32_t j = tid; j < remaining * ESZ; j += 256u) obase[body_bytes + j] = strm[tail_pos + j];

    // ---- D: samples = running sum of the deltas down each column (mod 2^W)
    auto delta_of = [](int x) -> uint32_t { return (W == 16 && !LOW) ? (uint32_t)x >> 16 : (uint32_t)x; };
    const uint32_t NBS = NB | 1u;                          // bsum[column][block], odd pitch (banks)
    // D1: the sum of every (block, column)'s eight deltas; lanes as in phase B (adjacent lanes = adjacent columns: no bank conflicts)
    for (uint32_t b = tid >> LOG2DP; b < nblk; b += (uint32_t)T) {
        if (col_ok) {
            const v4i* q = (const v4i*)(err + ((size_t)b * (uint32_t)D + (uint32_t)lane_d) * 8u);
            const v4i x0 = q[0], x1 = q[1];
            bsum[(uint32_t)lane_d * NBS + b] = delta_of(x0[0]) + delta_of(x0[1]) + delta_of(x0[2]) + delta_of(x0[3]) +
                                               delta_of(x1[0]) + delta_of(x1[1]) + delta_of(x1[2]) + delta_of(x1[3]);
        }
    }
    __syncthreads();
    // D2: exclusive prefix over the blocks of each column: T adjacent lanes a column, a contiguous piece each
    {
        const uint32_t d = tid >> LOG2T, t = tid & (uint32_t)(T - 1);
        const uint32_t per = (nblk + (uint32_t)T - 1u) >> LOG2T;
        const uint32_t i0 = t * per < nblk ? t * per : nblk, i1 = i0 + per < nblk ? i0 + per : nblk;
        uint32_t* const row = bsum + (d < (uint32_t)D ? d : 0u) * NBS;
        uint32_t s = 0;
        if (d < (uint32_t)D)
            for (uint32_t i = i0; i < i1; i++) s += row[i];
        uint32_t total;
        uint32_t run = group_scan<T>(s, (int)t, total);
        if (d < (uint32_t)D)
            for (uint32_t i = i0; i < i1; i++) { const uint32_t v = row[i]; row[i] = run; run += v; }
    }
    __syncthreads();
    LAT_STAMP();
    // D3: the samples, to the row-major image (over the stream: nobody reads it any more)
    {
        U* const img = (U*)strm;
        for (uint32_t b = tid >> LOG2DP; b < nblk; b += (uint32_t)T) {
            if (col_ok) {
                const v4i* q = (const v4i*)(err + ((size_t)b * (uint32_t)D + (uint32_t)lane_d) * 8u);
                const v4i x0 = q[0], x1 = q[1];
                const int x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
                uint32_t pv = bsum[(uint32_t)lane_d * NBS + b];
                U* const o = img + (size_t)b * blk_elems + (uint32_t)lane_d;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    pv += delta_of(x[i]);
                    o[(uint32_t)i * (uint32_t)D] = (U)pv;
                }
            }
        }
    }
    __syncthreads();
    LAT_STAMP();
    {
        const uint32_t n16 = body_bytes >> 4;
        for (uint32_t i = tid; i < n16; i += 256u) ((uint4*)obase)[i] = ((const uint4*)strm)[i];
        for (uint32_t j = (n16 << 4) + tid; j < body_bytes; j += 256u) obase[j] = strm[j];
    }
    LAT_STAMP();
#ifdef SPRINTZ_LAT_TIMING
    if (tid == 0 && a.rets) {
        uint64_t r = 0;
        if (a.nchunks == 2) {                              // pipeline view: A end, B wave 1 first round, B wave 1 end, C first block, C end (since the pipeline's start)
            const int ks[5] = {0, 1, 3, 6, 7};
            for (int k = 0; k < 5; k++) { const uint64_t dt = (dbg[ks[k]] - stamp[1]) >> 1; r |= (dt < 4095 ? dt : 4095) << (12 * k); }
        } else {
            for (int k = 0; k < 5; k++) { const uint64_t dt = (stamp[k + 1] - stamp[k]) >> 1; r |= (dt < 4095 ? dt : 4095) << (12 * k); }
        }
        a.rets[chunk] = (int64_t)r;
        return;
    }
#endif
    finish((int64_t)(nblk * blk_elems + remaining));
#undef LAT_STAMP
#undef LAT_DBG
}

}  // namespace sprintz

// decode_fast.h -- the decoder the headline benchmark runs (general row-major
// payload layout, one column per lane, D <= 64): an LDS-staged, prefetching,
// instruction-lean version of decode_kernel.h with identical stream semantics
// (sprintz_xff_rle.cpp:569-1179, sprintz_delta_rle.cpp:418-772).
//
// What bounded the first two versions (profiles/r1_*): ~10 global-load
// wave-instructions per 8-row block, each carrying 64 scattered 4-byte
// requests over 8 streams (~30 cycles of texture-addresser time apiece), and
// two dependent HBM/L2 round trips per block.  Here instead:
//   * INPUT.  A group (DP lanes = one chunk) reads its compressed stream
//     strictly sequentially, so it is read AHEAD of the parser: one 16-byte
//     global load per lane per refill (DP*16 = one "unit") is issued a whole
//     group (16 rows) before its bytes are parsed and parked in a per-group LDS
//     ring of 6 units (+ an apron that mirrors the ring head, so that a group's
//     reads never wrap).  Headers, run lengths and bit fields are then LDS reads
//     (aligned ds_read2_b32 + v_alignbyte_b32) -- no global load sits on the
//     parse critical path.
//   * FIELDS. rows are byte aligned, so the in-byte shift of a column is the
//     same for all 8 rows: field = v_bfe_u32(dword at row base, shift, nbits).
//   * FIRE.   with E = err << W:  delta = sbfe((prev_delta*coef + E), W, W)
//     == sext_W(err + ((prev_delta*coef) >> W)) exactly, W-bit wrap included
//     (the low W bits of the product cannot carry into a multiple of 2^W);
//     sign(err) for the coefficient gradient is one v_med3_i32.
//   * HEADER. both slots' nbits ride in one register through a DPP scan.
//   * OUTPUT. the 8 x D block is transposed through LDS, leaves as dwordx4.
#pragma once

#include "decode_kernel.h"
#include "group_ops.h"

namespace sprintz {

typedef uint32_t __attribute__((aligned(1), may_alias)) u32_unaligned;
typedef uint16_t __attribute__((aligned(1), may_alias)) u16_unaligned;
struct __attribute__((aligned(1), packed)) u128_unaligned { uint32_t x, y, z, w; };

typedef __attribute__((address_space(3))) const uint32_t lds_u32;
typedef __attribute__((address_space(3))) const uint8_t lds_u8;

// 32 bits at ANY byte address `a` of LDS.  gfx950 replays a misaligned
// ds_read_b32 (SQ_LDS_UNALIGNED_STALL: it made the first ring version 1.7x
// slower than the global-load kernel), so: one aligned ds_read2_b32 +
// v_alignbyte_b32, which takes the byte phase straight from the low two address
// bits (verified on hardware, tools/probes/alignbyte.hip).
__device__ __forceinline__ uint32_t lds_rd32(uint32_t a)
{
    lds_u32* q = (lds_u32*)(uintptr_t)(a & ~3u);
    return __builtin_amdgcn_alignbyte(q[1], q[0], a);
}
__device__ __forceinline__ uint32_t lds_rd8(uint32_t a) { return *(lds_u8*)(uintptr_t)a; }
__device__ __forceinline__ uint32_t lds_addr(const void* p)
{
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

// a 64-bit value that is the same in every lane of the wavefront, moved to SGPRs
// (the builtin returns int: go through uint32_t, or a low word >= 2^31 sign-extends into the high one)
__device__ __forceinline__ uint64_t wave_uniform64(uint64_t x)
{
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32)) << 32) |
           (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
}

// 24-bit multiply-add / multiply, pinned to the full-rate opcodes (hipcc turns
// __mul24 of values it cannot range-prove into quarter-rate v_mul_lo_u32)
__device__ __forceinline__ int mad24(int a, int b, int c)
{
    int d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// sign(x) in {-1, 0, 1} = clamp(x, -1, 1): one v_med3_i32 (hipcc lowers the C++
// min/max form to two cmp+cndmask pairs)
__device__ __forceinline__ int sign_of(int x)
{
    int s;
    asm("v_med3_i32 %0, %1, -1, 1" : "=v"(s) : "v"(x));
    return s;
}

// 16-bit FIRE step without the shift: D = (i16)hi16(a) * (i16)b + c.  With X = prev_delta*coef + E
// the new delta IS the high half of X, so the recurrence feeds X straight back in
// (verified against the C model and timed at full rate on hardware: tools/probes/mad_i16.hip).
__device__ __forceinline__ int mad_i16_hi(int a, int b, int c)
{
    int d;
    asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// a + hi16(x): the running value only matters modulo 2^16, so the unsigned high half will do
__device__ __forceinline__ uint32_t add_hi16(uint32_t a, int x)
{
    uint32_t d;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(d) : "v"(a), "v"(x));
    return d;
}

// CPL columns per lane (column = lane_d*CPL + k); EXACT: ndims == DP*CPL, so every
// size is a compile-time constant.
// CM: column-major destination (DecodeArgs::col_stride): the 8 samples a lane produces per block
// are contiguous in ITS column -- packed in registers and stored as one 16-byte (8-byte at
// W == 8) piece per column, no LDS transpose.
// DS != 0: the LDS carve (ring, apron, requests per step) is sized for streams of at most DS columns instead of the
// DP*CPL the lanes could hold (the launch checks ndims <= DS) -- what decides how many chunks a CU keeps in flight.
// CPL == 3 is the SPLIT mapping for 8-bit streams of 65 .. 96 columns on 32 lanes (two chunks a wavefront instead of one
// on 64 x 2 with 40 lanes busy at 80 columns): lane l owns the PAIR (2l, 2l+1) -- every pair genuine, on an even column,
// so the paired window / paired staging write of the 8-bit path apply -- and the SINGLE column 64 + l.
template <int W, bool FIRE, int DP, int CPL, bool EXACT, int Q = 0, bool CM = false, int DS = 0>
__global__ void __launch_bounds__(kThreads) decode_fast_kernel(DecodeArgs a)
{
    using U = typename Elem<W>::U;
    typedef __attribute__((__vector_size__(4 * sizeof(uint32_t)))) uint32_t v4u;
    constexpr int HB = Elem<W>::HB;
    constexpr int ESZ = W / 8;
    constexpr int LOG2DP = DP == 4 ? 2 : DP == 8 ? 3 : DP == 16 ? 4 : DP == 32 ? 5 : 6;
    constexpr int DCAP = DP * CPL;                         // columns a group can hold
    constexpr uint32_t ROW16 = DP * 16;                    // bytes one wave-instruction moves per group
    constexpr uint32_t UNIT = ROW16 * CPL;                 // bytes one refill unit brings in (CPL pieces per lane)
    constexpr bool SPLIT = CPL == 3;
    static_assert(!SPLIT || (W == 8 && DP == 32 && !EXACT && !CM && Q == 0), "the split mapping is built for 8-bit row-major decodes");
    constexpr int DSZ = DS ? DS : DCAP;                    // columns the LDS carve is sized for
    static_assert(DSZ <= DCAP, "sizing columns");
    constexpr uint32_t HDRMAX = (2 * DSZ * HB + 7) / 8;
    constexpr uint32_t BLKMAX = 8 * DSZ * ESZ;             // largest block payload = largest decoded block
    constexpr int PIECES = (BLKMAX + ROW16 - 1) / ROW16;   // 16-byte pieces of a decoded block per lane
    constexpr uint32_t CG = HDRMAX + 2 * BLKMAX + 4;       // most bytes one group can consume
    constexpr uint32_t NPEND = (CG + UNIT - 1) / UNIT;     // units requested per group step (2 or 3)
    constexpr uint32_t CSTART = 16 + 8;                    // chunk start: alignment gap + 8-byte stream header
    // (a ring of 4 units instead of 6 -- 960 bytes a group, 20 waves a CU instead of 16, safe only for compressible
    //  streams -- was timed on the headline batch: 0.4124 -> 0.4079 ms.  Occupancy is not what holds this kernel.)
    constexpr uint32_t RBU = (2 * (CG + CSTART) + 3 + UNIT - 1) / UNIT + 1;   // ring units: 6 @16 bit, 4 @8 bit
    constexpr uint32_t RB = RBU * UNIT;                    // ring bytes
    constexpr uint32_t APRON = (CG + CSTART + 8 + 15) & ~15u;   // a step never reads past its start + APRON
    static_assert(RB - UNIT >= 2 * (CG + CSTART) + 3, "ring too small for one step of read-ahead");
    static_assert(NPEND <= 3, "pending registers");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int D = EXACT ? DCAP : a.D;
    const uint64_t gtid = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
    const int lane_d = (int)(threadIdx.x & (uint32_t)(DP - 1));
    // A group decodes `chunks_per_group` CONSECUTIVE chunks.  Their streams are
    // (nearly) contiguous in the container, so the ring's read-ahead runs straight
    // across chunk boundaries: one cold start per group instead of one per chunk.
    const uint64_t c_first = (gtid >> LOG2DP) * (uint64_t)a.chunks_per_group;
    if (c_first >= a.nchunks) return;
    const uint64_t c_end = (c_first + a.chunks_per_group < a.nchunks) ? c_first + a.chunks_per_group : a.nchunks;

    // LDS carve per group: [ring RB | apron APRON | block staging]
    uint8_t* const ringp = smem + (size_t)(threadIdx.x >> LOG2DP) * a.lds_group_stride;
    const uint32_t ring = lds_addr(ringp);                 // LDS byte address of the ring
    uint8_t* const stage = ringp + RB + APRON;

    // stream geometry.  All stream loads go through ONE wave-uniform buffer
    // descriptor that spans the container from this wavefront's first stream to its
    // end: offsets are 32-bit, and a load that runs past the container returns 0
    // instead of faulting -- the read-ahead needs no bounds test.
    const uint64_t wave_first = (((uint64_t)blockIdx.x * kThreads + (threadIdx.x & ~63u)) >> LOG2DP) * (uint64_t)a.chunks_per_group;
    uint64_t wave_base = a.offsets[wave_first < a.nchunks ? wave_first : 0] & ~(uint64_t)15;
    wave_base = wave_uniform64(wave_base);
    // rounded up to whole 16-byte pieces (gfx950 zeroes a dwordx4 whose END is out of range);
    // the <= 15 extra bytes are inside the SPRINTZ_MI355X_READ_SLACK the API asks for
    const uint64_t wave_span = ((a.offsets[a.nchunks] - wave_base) + 15) & ~(uint64_t)15;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.comp + wave_base), 0, (uint32_t)(wave_span < 0xffffffffull ? wave_span : 0xffffffffull), 0x00020000);
    // output: one descriptor per wave as well, based at its first chunk's slot; a store
    // whose offset is out of range is dropped by the hardware, which is what the
    // unconditional per-step block stores rely on before a lane has produced a block
    // (column-major: the descriptor starts at this wave's first ROW of column 0 and runs to the
    //  end of the last column; offsets are column*col_stride + row, in bytes)
    const uint32_t rows_per_chunk = CM ? a.chunk_len / (uint32_t)(EXACT ? DCAP : a.D) : 0u;
    const uint64_t out_base = CM ? (wave_first < a.nchunks ? wave_first : 0) * (uint64_t)rows_per_chunk * ESZ
                                 : (wave_first < a.nchunks ? wave_first : 0) * (uint64_t)a.chunk_len * ESZ;
    const uint64_t out_span = CM ? (uint64_t)(EXACT ? DCAP : a.D) * a.col_stride * ESZ - out_base
                                 : a.nchunks * (uint64_t)a.chunk_len * ESZ - out_base;
    // (both are wave-uniform by construction; saying so keeps hipcc from wrapping every store in a
    //  readfirstlane waterfall loop -- 8 VALU per store it cannot prove away)
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((uint8_t*)a.out + wave_uniform64(out_base)), 0,
        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(out_span < 0xfffffff0ull ? out_span : 0xfffffff0ull)), 0x00020000);
    constexpr uint32_t kDropStore = 0xfffffff0u;          // out of range of every descriptor
    // the decoded samples are written once and not read again by this kernel: non-temporal stores (aux bit 1 = nt)
    // keep them from displacing the compressed streams in L2 and let whole lines go out -- measured with the
    // univariate decoder first (0.402 -> 0.336 ms on BASELINE config 1)
#ifndef SPRINTZ_STORE_AUX
#define SPRINTZ_STORE_AUX 2
#endif
#ifndef SPRINTZ_DF_MERGE8
#define SPRINTZ_DF_MERGE8 1
#endif
    constexpr int kStoreAux = SPRINTZ_STORE_AUX;
    const uint32_t lane16 = (uint32_t)lane_d * 16u;
    uint64_t gabs = 0;                                     // container offset the cursors below are relative to
    uint32_t gvo = 0;                                      // this lane's next 16 bytes to request (offset from wave_base)
    uint32_t rp = 0;                                       // parse cursor (offset from gabs)
    uint32_t rofs = 0;                                     // parse cursor (ring offset)
    uint32_t ahead = 0;                                    // bytes requested and not yet parsed
    uint32_t cofs = lane16;                                // ring offset where this lane parks its next 16 bytes

    uint4 pend[3][CPL];
    uint32_t npend = 0;

    // Request one unit (CPL 16-byte pieces per lane).  The loads are UNCONDITIONAL (an
    // unwanted unit asks for an offset outside the descriptor) so that the compiler can count
    // the VMEM operations of a step exactly -- see the wait discussion at the bottom of the loop.
    auto request = [&](uint4 (&v)[CPL], bool wanted) {
#pragma unroll
        for (int j = 0; j < CPL; j++) {
                        // (non-temporal LOADS were tried too: the read-ahead re-reads what neighbouring groups fetched, and with nt
            //  those re-reads go back to HBM -- 0.413 -> 0.461 ms)
            // an unwanted unit asks for an offset outside the descriptor: the texture addresser answers zeros without a request to
            // the cache (a re-read of offset 0 is a real request of 64 lanes, and the addresser's queue is what the block stores wait in)
            const auto t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, wanted ? gvo + j * ROW16 : 0xfffffff0u, 0, 0);
            v[j] = make_uint4(t[0], t[1], t[2], t[3]);
        }
        gvo += wanted ? UNIT : 0u;
    };
    auto commit = [&](const uint4 (&v)[CPL]) {             // park one unit in the ring (+ mirror the ring head)
#pragma unroll
        for (int j = 0; j < CPL; j++) {
            const uint32_t ro = cofs + j * ROW16;
            *(uint4*)(ringp + ro) = v[j];
            if (ro < APRON) *(uint4*)(ringp + RB + ro) = v[j];
        }
        cofs += UNIT;
        if (cofs >= RB) cofs -= RB;
    };
    // (re)start the read-ahead at container offset `off`: fill the whole ring
    auto prime = [&](uint64_t off) {
        gabs = off & ~(uint64_t)15;
        gvo = (uint32_t)(gabs - wave_base) + lane16;
        rp = (uint32_t)(off & 15);
        rofs = rp;
        cofs = lane16;
        npend = 0;
        wave_lds_sync();
#pragma unroll
        for (uint32_t u = 0; u < RB / UNIT; u++) {
            uint4 v[CPL];
            request(v, true);
            commit(v);
        }
        ahead = RB - rp;
        wave_lds_sync();
    };

    const uint32_t hdr_bytes = (2u * (uint32_t)D * HB + 7u) >> 3;
    const uint32_t blk_elems = 8u * (uint32_t)D;
    const uint32_t blk_bytes = blk_elems * ESZ;
    const uint32_t row_stride = (uint32_t)D * ESZ;
    const int col0 = lane_d * CPL;                         // first column of this lane
    // A lane column past the last one (ndims is not DP*CPL) stands in for column D-1: it reads the same header fields and
    // bit fields, runs the same recurrence and stages the same bytes at the same place as the genuine one -- so the per-sample
    // code needs no predicate (54 exec-mask regions a group step at 80 columns on 64 x 2); only the scan must not count it.
    // (8 bits, an even number of columns per lane, row-major output: a lane's two ADJACENT columns leave as one 16-bit staging
    //  write per row -- half the ds_write instructions, which is what bounds this shape (4 LDS cycles per ds_write_b8, 2 per
    //  sample for the field window).  A lane past the last column then stands in for the last PAIR, (D-2, D-1) -- the same two
    //  bytes at the same address as the genuine lane's; with an odd D the last lane's second byte spills into the next row's
    //  first, which that row's own write -- DS operations of a wave execute in order -- puts right, and after row 7 into padding.)
    constexpr bool MERGE8 = SPRINTZ_DF_MERGE8 && W == 8 && (CPL % 2 == 0 || SPLIT) && !CM && Q == 0;
    const bool merge8 = MERGE8 && (EXACT || (D & 1) == 0);   // rows of an odd number of bytes would make every second 16-bit write misaligned
    constexpr int NPAIR = CPL / 2;                         // columns 2j, 2j+1 (j < NPAIR) of a lane are adjacent in the row
    int genk[CPL];                                         // the column this lane column IS (>= D: it is a stand-in)
    int colk[CPL];
    uint8_t* stage_k[CPL];
#pragma unroll
    for (int k = 0; k < CPL; k++) {
        genk[k] = SPLIT ? (k < 2 ? 2 * lane_d + k : 2 * DP + lane_d) : col0 + k;
        if (MERGE8 && merge8 && !SPLIT) {
            const int pb = (col0 + (k & ~1)) < D ? col0 + (k & ~1) : ((D - 1) & ~1);      // first column of this lane's pair
            colk[k] = EXACT ? col0 + k : (pb + (k & 1) < D ? pb + (k & 1) : D - 1);
        } else {
            colk[k] = EXACT ? genk[k] : (genk[k] < D ? genk[k] : D - 1);
        }
        stage_k[k] = stage + colk[k] * ESZ;
    }

    uint32_t pv[CPL];
    int pd[CPL], ctr[CPL];
    bool col_ok[CPL];
#pragma unroll
    for (int k = 0; k < CPL; k++) { pv[k] = 0; pd[k] = 0; ctr[k] = 0; col_ok[k] = EXACT ? true : genk[k] < D; }
    uint32_t out_left = 0;                                 // capacity guard (elements)
    bool corrupt = false;
    // query-on-compressed (Q != 0): per-column max and sum of the chunk, kept next to the
    // predictor state; with Q == kQueryReduceOnly the block never leaves the registers
    uint32_t qmax[CPL];
    uint64_t qsum[CPL];
    uint32_t qbs[CPL];
#pragma unroll
    for (int k = 0; k < CPL; k++) { qmax[k] = 0; qsum[k] = 0; qbs[k] = 0; }
    auto q_row = [&](int k) {                              // pv[k] carries garbage above bit W: SDWA selects the element
        if constexpr (Q != 0) {
            if constexpr (W == 16) {
                asm("v_max_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"
                    : "=v"(qmax[k]) : "v"(qmax[k]), "v"(pv[k]));
                asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"
                    : "=v"(qbs[k]) : "v"(qbs[k]), "v"(pv[k]));
            } else {
                asm("v_max_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0"
                    : "=v"(qmax[k]) : "v"(qmax[k]), "v"(pv[k]));
                asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0"
                    : "=v"(qbs[k]) : "v"(qbs[k]), "v"(pv[k]));
            }
        }
    };
    auto q_block = [&](int k) {
        if constexpr (Q != 0) { qsum[k] += qbs[k]; qbs[k] = 0; }
    };
    uint32_t ovo = 0;                                      // output cursor (byte offset from this wave's out_base)

    // ---- per-block workers ------------------------------------------------------
    // The staged 8 x D block is contiguous in the output.  Packed blocks are read
    // back from LDS into `held[slot]` and stored at the BOTTOM of the group step by
    // unconditional dwordx4 buffer stores (a slot that produced no packed block
    // re-stores the previous one: same lane, same address, same data; before the
    // first block the offset is out of range and the hardware drops the store).
    uint4 held[2][PIECES];
    uint32_t held_vo[2][PIECES];
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
        for (int q = 0; q < PIECES; q++) { held[s2][q] = make_uint4(0, 0, 0, 0); held_vo[s2][q] = kDropStore; }
    // after the 8 rows sit in `stage`; slot < 0: store right away (run blocks)
    // column-major: the packed 8 samples of each of this lane's columns
    uint32_t pk[CPL][4];
    uint4 cheld[2][CPL];
    uint32_t cheld_vo[2][CPL];
    uint32_t cbase[CPL];                                   // byte offset of this lane's columns in the descriptor
#pragma unroll
    for (int k = 0; k < CPL; k++) {
        cbase[k] = CM ? (uint32_t)((uint64_t)genk[k] * a.col_stride * ESZ) : 0u;
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) { cheld[s2][k] = make_uint4(0, 0, 0, 0); cheld_vo[s2][k] = kDropStore; }
    }
    auto store_col = [&](const uint4& t, uint32_t vo) {
        if constexpr (W == 16) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, t), orsrc, vo, 0, kStoreAux);
        } else {
            typedef __attribute__((__vector_size__(2 * sizeof(uint32_t)))) uint32_t v2u;
            v2u h = {t.x, t.y};
            __builtin_amdgcn_raw_buffer_store_b64(h, orsrc, vo, 0, kStoreAux);
        }
    };
    // Column-major burst (one column per lane): a lone 16-byte store per lane per block costs
    // more than the whole decode (0.36 vs 0.15 ms on 8M x 32; every request is its own cache
    // line).  So four blocks (two group steps) of every column wait in LDS as
    // cst[column][slot], and on every second step a QUAD of lanes stores one column's 64
    // contiguous bytes -- whole 64-byte requests, as in the row-major path.
    constexpr bool CMB = CM && CPL == 1;
    constexpr uint32_t PB = W == 16 ? 16u : 8u;            // bytes of one block of one column
    uint8_t* const cst = stage;
    uint32_t* const cvo = (uint32_t*)(stage + 4u * DCAP * PB);   // row offset of staged block b, or kDropStore
    uint32_t stepno = 0;
    const uint32_t cs_bytes = CM ? (uint32_t)(a.col_stride * ESZ) : 0u;
    // slot permutation: column c keeps block b at slot (b + c/4) % 4, which spreads both the
    // per-column writes and the per-quad reads over the banks
    auto cst_at = [&](uint32_t col, uint32_t blk) { return cst + (col * 4u + ((blk + (col >> 2)) & 3u)) * PB; };
    if constexpr (CMB) {
        if (lane_d < 4) cvo[lane_d] = kDropStore;
        wave_lds_sync();
    }
    // `real` = false: the same four store instructions, all dropped (keeps the VMEM count of a step fixed)
    auto cm_flush = [&](bool real) {
        if constexpr (CMB) {
            wave_lds_sync();
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t pid = (uint32_t)j * DP + (uint32_t)lane_d, col = pid >> 2, p = pid & 3u;
                uint4 t = make_uint4(0, 0, 0, 0);
                uint32_t vo = kDropStore;
                if (real && col < (uint32_t)D) {
                    const uint32_t rofs_b = cvo[p];
                    if constexpr (W == 16) t = *(const uint4*)cst_at(col, p);
                    else { const uint2 h = *(const uint2*)cst_at(col, p); t.x = h.x; t.y = h.y; }
                    vo = rofs_b == kDropStore ? kDropStore : col * cs_bytes + rofs_b;
                }
                store_col(t, vo);
            }
            if (real) {
                wave_lds_sync();
                if (lane_d < 4) cvo[lane_d] = kDropStore;
            }
            wave_lds_sync();
        }
    };
    auto pack_row = [&](int k, int i) {                    // pv[k] is row i of the block
        if constexpr (CM) {
            if constexpr (W == 16) {
                if ((i & 1) == 0) pk[k][i >> 1] = pv[k] & 0xffffu;
                else pk[k][i >> 1] |= pv[k] << 16;
            } else {
                if ((i & 3) == 0) pk[k][i >> 2] = pv[k] & 0xffu;
                else pk[k][i >> 2] |= (pv[k] & 0xffu) << (8 * (i & 3));
            }
        }
    };
    auto stage_out = [&](int slot) {
        if constexpr (Q == kQueryReduceOnly) return;
        if constexpr (CMB) {                               // ovo counts ROW bytes here
            if (slot >= 0) {
                const uint32_t blk = (stepno & 1u) * 2u + (uint32_t)slot;
                if (col_ok[0]) {
                    if constexpr (W == 16) *(uint4*)cst_at((uint32_t)col0, blk) = make_uint4(pk[0][0], pk[0][1], pk[0][2], pk[0][3]);
                    else *(uint2*)cst_at((uint32_t)col0, blk) = make_uint2(pk[0][0], pk[0][1]);
                }
                if (lane_d == 0) cvo[blk] = ovo;
            } else {                                       // run blocks: straight from the registers
                const uint4 t = W == 16 ? make_uint4(pk[0][0], pk[0][1], pk[0][2], pk[0][3]) : make_uint4(pk[0][0], pk[0][1], 0, 0);
                store_col(t, col_ok[0] ? cbase[0] + ovo : kDropStore);
            }
            ovo += 8u * ESZ;
            return;
        }
        if constexpr (CM) {
#pragma unroll
            for (int k = 0; k < CPL; k++) {
                const uint4 t = W == 16 ? make_uint4(pk[k][0], pk[k][1], pk[k][2], pk[k][3]) : make_uint4(pk[k][0], pk[k][1], 0, 0);
                const uint32_t vo = col_ok[k] ? cbase[k] + ovo : kDropStore;
                if (slot >= 0) { cheld[slot][k] = t; cheld_vo[slot][k] = vo; }
                else store_col(t, vo);
            }
            ovo += 8u * ESZ;
            return;
        }
        wave_lds_sync();
#pragma unroll
        for (int q = 0; q < PIECES; q++) {
            const uint32_t u = lane16 + q * ROW16;
            const bool in = u < blk_bytes;
            const uint4 t = *(const uint4*)(stage + (in ? u : 0u));
            if (slot >= 0) {
                held[slot][q] = t;
#ifdef SPRINTZ_ABL_NO_GSTORE
                held_vo[slot][q] = kDropStore;             // ablation: every block store is dropped by the descriptor
#elif defined(SPRINTZ_ABL_STORE_INTERLEAVE)
                {   // ablation (wrong output on purpose): the 8 groups of a wave write ONE contiguous kilobyte per store instruction
                    const uint32_t g = (threadIdx.x & 63u) >> LOG2DP, gbase = g * a.chunk_len * ESZ;
                    held_vo[slot][q] = in ? ((ovo - gbase) >> 7) * 1024u + g * 128u + u : kDropStore;
                }
#elif defined(SPRINTZ_ABL_STORE_SLOT0_ONLY)
                held_vo[slot][q] = (in && slot == 0) ? ovo + u : kDropStore;   // ablation: slot 1's store is dropped
#elif defined(SPRINTZ_ABL_STORE_HALF_LANES)
                held_vo[slot][q] = (in && (lane_d & 1) == 0) ? ovo + u : kDropStore;   // ablation: every second lane's 16 bytes are dropped
#else
                held_vo[slot][q] = in ? ovo + u : kDropStore;
#endif
            } else {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, t), orsrc, in ? ovo + u : kDropStore, 0, kStoreAux);
            }
        }
        wave_lds_sync();
        ovo += blk_bytes;
    };
    auto run_blocks = [&](uint32_t len) {                  // RUN slot: `len` blocks of zero error (:828-958)
        for (; len > 0; len--) {
            if (out_left < blk_elems) { corrupt = true; break; }
            out_left -= blk_elems;
            auto run_step = [&](int k, int coef) {
                if constexpr (W == 16 && FIRE) {            // pd[k] holds X (delta in its high half), see packed_block
                    pd[k] = mad_i16_hi(pd[k], coef, 0);
                    pv[k] = add_hi16(pv[k], pd[k]);
                } else {
                    const int delta = FIRE ? __builtin_amdgcn_sbfe(mad24(pd[k], coef, 0), W, W) : 0;
                    pv[k] += (uint32_t)delta;
                    pd[k] = delta;
                }
            };
            auto run_col = [&](int k) {
                int coef = FIRE ? fire_coef<W, false>(ctr[k]) : 0;
                if constexpr (FIRE && W == 16) {
                    if (a.quirk) coef = fire_coef_ref_run16(ctr[k], genk[k]);      // the reference decoder's run replay, on request
                }
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    run_step(k, coef);
                    q_row(k);
                    pack_row(k, i);
#ifndef SPRINTZ_ABL_NO_STAGE
                    if constexpr (Q != kQueryReduceOnly && !CM)
                        *(U*)(stage_k[k] + i * row_stride) = (U)pv[k];
#endif
                }
                q_block(k);
            };
            if (MERGE8 && merge8) {
#pragma unroll
                for (int j = 0; j < NPAIR; j++) {
                    const int k = 2 * j;
                    const int coef0 = FIRE ? fire_coef<W, false>(ctr[k]) : 0, coef1 = FIRE ? fire_coef<W, false>(ctr[k + 1]) : 0;
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        run_step(k, coef0);
                        run_step(k + 1, coef1);
                        *(uint16_t*)(stage_k[k] + i * row_stride) = (uint16_t)__builtin_amdgcn_perm(pv[k + 1], pv[k], 0x0c0c0400u);
                    }
                }
                if constexpr (CPL & 1) run_col(CPL - 1);
            } else {
#pragma unroll
                for (int k = 0; k < CPL; k++) run_col(k);
            }
            stage_out(-1);
        }
    };
    // Field fetch fused with zigzag^-1.  With z = bits [sh, sh+nb) of the loaded dword,
    // err = (z >> 1) ^ -(z & 1) = bfe_u(w, sh+1, nb-1) ^ bfe_i(w, sh, 1): both halves come
    // straight from the dword (3 VALU per sample instead of bfe + 3); for an empty column
    // (nb == 0) both widths are 0 and the field reads as 0.  The XOR lands in the UPPER
    // half of the register (SDWA dst_sel:WORD_1), i.e. it yields E = err << 16 directly,
    // which is what the FIRE step and the sign test want (W == 16); for W == 8 one shift follows (FIRE) or none (delta).
    auto fetch_rows = [&](int (&e)[CPL][8], uint32_t at, const uint32_t (&off)[CPL], const uint32_t (&nb)[CPL], uint32_t row_bytes) {
        if constexpr (W == 8 && CPL >= 2) {
            // 8 bits: a lane's two adjacent columns are two adjacent fields of at most 8 bits, at most 7 + 16 bits from the byte
            // the first one starts in -- ONE 32-bit window per row serves both (an address, a mask and a v_alignbyte less per
            // second column and row).  (A stand-in column repeats the last genuine one: the same window.)
#pragma unroll
            for (int j = 0; j < NPAIR; j++) {
                const int k = 2 * j;
                uint32_t p = at + (off[k] >> 3);
                const uint32_t sha = off[k] & 7u, shb = off[k + 1] - (off[k] & ~7u);
                const uint32_t w1a = nb[k] != 0 ? 1u : 0u, wma = nb[k] - w1a;
                const uint32_t w1b = nb[k + 1] != 0 ? 1u : 0u, wmb = nb[k + 1] - w1b;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const uint32_t w = lds_rd32(p);
                    const uint32_t ea = __builtin_amdgcn_ubfe(w, sha + 1u, wma) ^ (uint32_t)__builtin_amdgcn_sbfe((int)w, sha, w1a);
                    const uint32_t eb = __builtin_amdgcn_ubfe(w, shb + 1u, wmb) ^ (uint32_t)__builtin_amdgcn_sbfe((int)w, shb, w1b);
                    e[k][i] = FIRE ? (int)ea << 8 : (int)ea;
                    e[k + 1][i] = FIRE ? (int)eb << 8 : (int)eb;
                    p += row_bytes;
                }
            }
            if constexpr (CPL % 2 == 0) return;
        }
#pragma unroll
        for (int k = (W == 8 && CPL >= 2) ? 2 * NPAIR : 0; k < CPL; k++) {
            uint32_t p = at + (off[k] >> 3);
            const uint32_t sh = off[k] & 7u;
            const uint32_t w1 = nb[k] != 0 ? 1u : 0u;      // width of the sign bit field
            const uint32_t wm = nb[k] - w1;                // width of the magnitude field
#pragma unroll
            for (int i = 0; i < 8; i++) {
#ifdef SPRINTZ_ABL_NO_FETCH
                const uint32_t w = p * 2654435761u;        // ablation: no LDS read, the address arithmetic stays
#else
                const uint32_t w = lds_rd32(p);
#endif
                const uint32_t mag = __builtin_amdgcn_ubfe(w, sh + 1u, wm);
                const int sgn = __builtin_amdgcn_sbfe((int)w, sh, w1);
                if constexpr (W == 16) {
                    int x;
                    asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0"
                        : "=v"(x) : "v"(mag), "v"(sgn));
                    e[k][i] = x;                           // err << 16
                } else if constexpr (FIRE) {
                    e[k][i] = (int)(mag ^ (uint32_t)sgn) << 8;            // err << 8 (mag ^ sgn is err, sign and all)
                } else {
                    e[k][i] = (int)(mag ^ (uint32_t)sgn);                 // the delta itself: nothing to scale it for
                }
                p += row_bytes;
            }
        }
    };
    // zigzag^-1 is done; e[][] holds E = err << W (sign-extended to 32 bits); 8-bit delta coding: err itself
    auto packed_block = [&](const int (&e)[CPL][8], int slot) {   // forecast recurrence (:993-1150)
        if (out_left < blk_elems) { corrupt = true; return; }
        out_left -= blk_elems;
        auto col_step = [&](int k, int i, int coef, int& grad) {
            if constexpr (W == 16 && FIRE) {
                // X = prev_delta*coef + E; delta = hi16(X): pd[k] carries X, never the shifted delta
                if (i & 1) grad = mad_i16_hi(pd[k], sign_of(e[k][i]), grad);   // sign(E) == sign(err)
                pd[k] = mad_i16_hi(pd[k], coef, e[k][i]);
                pv[k] = add_hi16(pv[k], pd[k]);
            } else if constexpr (W == 16) {
                pv[k] = add_hi16(pv[k], e[k][i]);                               // delta = E >> 16
            } else {
                int delta;
                if constexpr (FIRE) {
                    if (i & 1) grad = mad24(sign_of(e[k][i]), pd[k], grad);
                    delta = __builtin_amdgcn_sbfe(mad24(pd[k], coef, e[k][i]), W, W);
                } else {
                    delta = e[k][i];
                }
                pv[k] += (uint32_t)delta;
                pd[k] = delta;
            }
        };
        auto one_col = [&](int k) {
            int grad = 0;
            const int coef = FIRE ? fire_coef<W, false>(ctr[k]) : 0;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                col_step(k, i, coef, grad);
                q_row(k);
                pack_row(k, i);
#ifndef SPRINTZ_ABL_NO_STAGE
                if constexpr (Q != kQueryReduceOnly && !CM)
                    *(U*)(stage_k[k] + i * row_stride) = (U)pv[k];
#else
                asm volatile("" :: "v"(pv[k]));
#endif
            }
            q_block(k);
            if constexpr (FIRE) ctr[k] = wrap_counter<W>(ctr[k] + __builtin_amdgcn_sbfe(grad, 2, W - 2));   // sext_W(grad) >> 2
        };
        if (MERGE8 && merge8) {                            // a lane's two adjacent 8-bit columns: one 16-bit staging write per row
#pragma unroll
            for (int j = 0; j < NPAIR; j++) {
                const int k = 2 * j;
                int grad0 = 0, grad1 = 0;
                const int coef0 = FIRE ? fire_coef<W, false>(ctr[k]) : 0, coef1 = FIRE ? fire_coef<W, false>(ctr[k + 1]) : 0;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    col_step(k, i, coef0, grad0);
                    col_step(k + 1, i, coef1, grad1);
                    *(uint16_t*)(stage_k[k] + i * row_stride) = (uint16_t)__builtin_amdgcn_perm(pv[k + 1], pv[k], 0x0c0c0400u);
                }
                if constexpr (FIRE) {
                    ctr[k] = wrap_counter<W>(ctr[k] + __builtin_amdgcn_sbfe(grad0, 2, W - 2));
                    ctr[k + 1] = wrap_counter<W>(ctr[k + 1] + __builtin_amdgcn_sbfe(grad1, 2, W - 2));
                }
            }
            if constexpr (CPL & 1) one_col(CPL - 1);
        } else {
#pragma unroll
            for (int k = 0; k < CPL; k++) one_col(k);
        }
        stage_out(slot);
    };
    auto run_length = [&](uint32_t at, uint32_t& nbytes) -> uint32_t {   // varint in blocks (:829-833)
        const uint32_t b0 = lds_rd8(at);
        uint32_t len = b0 & 0x7fu;
        nbytes = 1;
        if (b0 & 0x80u) { len |= lds_rd8(at + 1) << 7; nbytes = 2; }
        return len;
    };

    bool need_prime = true;
    for (uint64_t chunk = c_first; chunk < c_end; chunk++) {
    // ---- chunk start: find the stream, read its 8-byte header (format.h:48-62)
    const uint64_t off_c = a.offsets[chunk];
    {
        const uint64_t off = off_c;
        const uint64_t gap = off - (gabs + rp);            // bytes between the cursor and the next stream
        if (need_prime || gap > 16 || rp > (1u << 30)) {
            prime(off);
        } else {                                           // alignment padding: just step over it
            rp += (uint32_t)gap;
            rofs += (uint32_t)gap;
            if (rofs >= RB) rofs -= RB;
            ahead -= (uint32_t)gap;
        }
        need_prime = false;
    }
    uint32_t groups_left, remaining;
    {
        const uint32_t w0 = lds_rd32(ring + rofs), w1 = lds_rd32(ring + rofs + 4);
        uint32_t hbytes = 8, nd_hdr = w1 >> 16;
        groups_left = w0;
        remaining = w1 & 0xffffu;
        if (a.norle) {                                     // {u32 len; u16 ndims} / u64 len with ndims in bytes 6..7 (decode_kernel.h)
            const uint32_t len = w0 <= a.chunk_len ? w0 : 0u;
            nd_hdr = w0 <= a.chunk_len ? (a.norle == 2 ? w1 >> 16 : w1 & 0xffffu) : 0xffffffffu;
            hbytes = a.norle == 2 ? 8u : 6u;
            groups_left = len < 128u ? 0u : len / (16u * (uint32_t)D);
            remaining = len - groups_left * 16u * (uint32_t)D;
        }
        rp += hbytes;
        rofs += hbytes;
        if (rofs >= RB) rofs -= RB;
        ahead -= hbytes;
#pragma unroll
        for (int k = 0; k < CPL; k++) { pv[k] = 0; pd[k] = 0; ctr[k] = 0; qmax[k] = 0; qsum[k] = 0; }
        out_left = a.chunk_len;
        ovo = CM ? (uint32_t)((chunk - wave_first) * (uint64_t)rows_per_chunk * ESZ)
                 : (uint32_t)((chunk - wave_first) * (uint64_t)a.chunk_len * ESZ);
        corrupt = (int)nd_hdr != D;
        // a damaged header must not make the loop spin: every group takes at least its
        // header and two slot bytes (none in the run-less codecs) out of the stream
        const uint64_t stream_len = a.offsets[chunk + 1] - off_c;
        if ((uint64_t)groups_left * (hdr_bytes + (a.norle ? 0u : 2u)) > stream_len || groups_left > a.chunk_len / blk_elems + 2u) corrupt = true;
        if (corrupt) groups_left = 0;
    }

    while (groups_left > 0 && !corrupt) {
        groups_left--;
        // ---- request the units that fit now; they are parked at the bottom of this step
        {
            const uint32_t room = RB - ahead;              // ring bytes the parser no longer needs
            {
            npend = 0;
#pragma unroll
            for (uint32_t k = 0; k < NPEND; k++) {
                const bool wanted = room >= (k + 1) * UNIT;
                request(pend[k], wanted);
                npend += wanted ? 1u : 0u;
            }
            ahead += npend * UNIT;
            }
        }

        // ---- group header: 2*D fields of HB bits (sprintz_xff_rle.cpp:713-735); both slots'
        // nbits ride in one register (slot 0 in bits 0..15, slot 1 in 16..31)
        const uint32_t r = ring + rofs;                    // LDS address of the parse cursor
        uint32_t nb_both[CPL], lane_both = 0;
        uint32_t hw0 = 0, hw1 = 0;                         // (two columns of a lane: their header fields sit in one window per slot)
#pragma unroll
        for (int k = 0; k < CPL; k++) {
            const uint32_t hbit0 = (uint32_t)colk[k] * HB, hbit1 = (uint32_t)(D + colk[k]) * HB;
            uint32_t s0 = hbit0 & 7u, s1 = hbit1 & 7u;
            if (k & 1) {                                   // the odd column reads its pair's window
                const uint32_t b0 = (uint32_t)colk[k - 1] * HB, b1 = (uint32_t)(D + colk[k - 1]) * HB;
                s0 = hbit0 - (b0 & ~7u);
                s1 = hbit1 - (b1 & ~7u);
            } else {
                hw0 = lds_rd32(r + (hbit0 >> 3));
                hw1 = lds_rd32(r + (hbit1 >> 3));
            }
            uint32_t f0 = __builtin_amdgcn_ubfe(hw0, s0, HB);
            uint32_t f1 = __builtin_amdgcn_ubfe(hw1, s1, HB);
            f0 += (f0 == (uint32_t)(W - 1));               // W-1 means W (:747-749)
            f1 += (f1 == (uint32_t)(W - 1));
            nb_both[k] = f0 | (f1 << 16);
            lane_both += col_ok[k] ? nb_both[k] : 0u;
        }
        uint32_t tot_both;
        uint32_t excl_both, excl_single = 0;
        if constexpr (SPLIT) {                             // columns 0 .. 63 are the pairs, lane by lane; the singles follow them
            uint32_t tot_pairs, tot_singles;
            excl_both = group_scan<DP>(nb_both[0] + nb_both[1], lane_d, tot_pairs);
            excl_single = tot_pairs + group_scan<DP>(col_ok[2] ? nb_both[2] : 0u, lane_d, tot_singles);
            tot_both = tot_pairs + tot_singles;
        } else {
            excl_both = group_scan<DP>(lane_both, lane_d, tot_both);
        }
        const uint32_t tot0 = tot_both & 0xffffu, tot1 = tot_both >> 16;
        uint32_t off0[CPL], off1[CPL], nb0[CPL], nb1[CPL];
#pragma unroll
        for (int k = 0; k < CPL; k++) {
            uint32_t o_both = col_ok[k] ? (SPLIT && k == 2 ? excl_single : excl_both) : tot_both - nb_both[k];   // (a stand-in: where column D-1 starts)
            if constexpr (MERGE8 && !EXACT && !SPLIT) {    // (a stand-in for column D-2 -- the first of the last pair -- starts one field earlier)
                if ((k & 1) == 0 && !col_ok[k] && colk[k] != colk[k + 1]) o_both -= nb_both[k + 1];
            }
            off0[k] = o_both & 0xffffu;
            off1[k] = o_both >> 16;
            nb0[k] = nb_both[k] & 0xffffu;
            nb1[k] = nb_both[k] >> 16;
            excl_both += nb_both[k];
        }

        // ---- lay out both slots, then fetch both before computing either
        const uint32_t at0 = r + hdr_bytes;
        uint32_t len0 = 0, len1 = 0, bytes0, bytes1;
        const uint32_t rb0 = (tot0 + 7u) >> 3, rb1 = (tot1 + 7u) >> 3;
        // (run-less codecs: an all-zero block has no payload at all and stands for itself)
        if (tot0 == 0) { if (a.norle) { len0 = 1; bytes0 = 0; } else len0 = run_length(at0, bytes0); } else bytes0 = rb0 * 8u;
        const uint32_t at1 = at0 + bytes0;
        if (tot1 == 0) { if (a.norle) { len1 = 1; bytes1 = 0; } else len1 = run_length(at1, bytes1); } else bytes1 = rb1 * 8u;
        const uint32_t used = hdr_bytes + bytes0 + bytes1;
        int z0[CPL][8], z1[CPL][8];
        if (tot0 != 0) fetch_rows(z0, at0, off0, nb0, rb0);
        if (tot1 != 0) fetch_rows(z1, at1, off1, nb1, rb1);

        if (tot0 == 0) run_blocks(len0); else packed_block(z0, 0);
        if (!corrupt) { if (tot1 == 0) run_blocks(len1); else packed_block(z1, 1); }

        rp += used;
        rofs += used;
        if (rofs >= RB) rofs -= RB;
        ahead -= used;

        // ---- bottom of the step.  VMEM order inside one step is: the stream loads (top),
        // [rare: run-block stores], the block stores (here).  gfx950 has ONE in-order
        // counter for loads and stores, so parking the loads must not wait for the stores
        // just issued: with every VMEM op of the common path unconditional, hipcc emits
        // s_waitcnt vmcnt(<number of stores>) here instead of vmcnt(0).
        if constexpr (Q != kQueryReduceOnly && CMB) {
            cm_flush((stepno & 1u) != 0);
            stepno++;
        } else if constexpr (Q != kQueryReduceOnly && CM) {
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
                for (int k = 0; k < CPL; k++) store_col(cheld[s2][k], cheld_vo[s2][k]);
        } else if constexpr (Q != kQueryReduceOnly) {
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
                for (int q = 0; q < PIECES; q++)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, held[s2][q]), orsrc, held_vo[s2][q], 0, kStoreAux);
        }
#pragma unroll
        for (uint32_t k = 0; k < NPEND; k++)
            if (k < npend) commit(pend[k]);
        wave_lds_sync();
    }

    if constexpr (CMB) {                                   // an odd number of steps leaves one step's blocks staged
        if (stepno & 1u) cm_flush(true);
        stepno = 0;
    }
    // ---- verbatim tail (:1171), straight from HBM
    const uint32_t out_elems = a.chunk_len - out_left;
    if (!corrupt && remaining > out_left) corrupt = true;
    if constexpr (Q != 0) {
        // the verbatim tail continues the row-major order: element e sits in column e % D
        if (!corrupt) {
            const uint8_t* t = a.comp + gabs + rp;
#pragma unroll
            for (int k = 0; k < CPL; k++) {
                if (!col_ok[k]) continue;
                for (uint32_t e = (uint32_t)genk[k]; e < remaining; e += (uint32_t)D) {
                    const uint32_t x = ESZ == 1 ? (uint32_t)t[e] : (uint32_t)*(const u16_unaligned*)(t + 2 * e);
                    qmax[k] = x > qmax[k] ? x : qmax[k];
                    qsum[k] += x;
                }
                if (a.qres) a.qres[chunk * (uint64_t)D + (uint64_t)genk[k]] = a.qop == 1 ? (uint64_t)qmax[k] : qsum[k];
            }
        }
    }
    if (!corrupt && remaining > 0 && Q != kQueryReduceOnly && CM) {
        const uint8_t* t = a.comp + gabs + rp;
        U* const c0 = (U*)((uint8_t*)a.out + out_base + ovo);          // column 0 at the tail's first row
        for (uint32_t e = (uint32_t)lane_d; e < remaining; e += DP) {
            const uint32_t x = ESZ == 1 ? (uint32_t)t[e] : (uint32_t)*(const u16_unaligned*)(t + 2 * e);
            c0[(uint64_t)(e % (uint32_t)D) * a.col_stride + e / (uint32_t)D] = (U)x;
        }
    } else if (!corrupt && remaining > 0 && Q != kQueryReduceOnly) {
        const uint8_t* t = a.comp + gabs + rp;
        uint8_t* d = (uint8_t*)a.out + out_base + ovo;
        copy_verbatim(t, d, remaining * ESZ, (uint32_t)lane_d, (uint32_t)DP);
    }
    if (corrupt || remaining > 0) need_prime = true;       // cursor no longer at the next stream
    if (lane_d == 0 && a.rets) a.rets[chunk] = corrupt ? kErrCorrupt : (int64_t)out_elems + remaining;
    }   // chunk loop
}

// bytes of LDS one group needs in decode_fast_kernel
constexpr uint32_t decode_fast_lds_bytes(int W, int DP, int CPL, int D, bool colmajor_burst = false, int DS = 0)
{
    const uint32_t unit = DP * 16 * CPL;
    const uint32_t hb = W == 8 ? 3 : 4;
    const uint32_t dcap = DS ? DS : DP * CPL;
    const uint32_t hdrmax = (2 * dcap * hb + 7) / 8, blkmax = 8 * dcap * (W / 8);
    const uint32_t cg = hdrmax + 2 * blkmax + 4;
    const uint32_t rb = ((2 * (cg + 24) + 3 + unit - 1) / unit + 1) * unit;
    const uint32_t apron = (cg + 24 + 8 + 15) & ~15u;
    // column-major burst staging: 4 blocks of every column the group can hold + 4 row offsets
    const uint32_t stage = colmajor_burst ? 4u * blkmax + 16 : ((8u * D * (W / 8) + 15) & ~15u) + 16;   // +16: spread groups over banks
    return rb + apron + stage;
}

}  // namespace sprintz

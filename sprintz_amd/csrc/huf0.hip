// huf0.hip -- batched decoder for genuine Huff0 blocks (Yann Collet's Huff0 / FSE as shipped in
// zstd 1.4.x: HUF_compress's output = tree description + jump table + 4 bit streams), one block
// per chunk -- the entropy stage the paper applies after bit-packing (communicate/ubicomp/
// method.tex:293-297) in ITS wire format, where huf.hip is this repository's own GPU-shaped
// container.  The format is restated, with the library citations, in oracle/huf0_oracle.c;
// parity is pinned against the system libzstd's HUF_compress / HUF_decompress by the tests.
//
// A Huff0 block is 4-way parallel by construction and its code is private to the block: a wave
// takes 16 chunks, lane = (chunk, stream).  The library's decoder looks codes up in a
// 2^tableLog-entry table; 16 such tables are 64 KB of LDS and leave a CU two waves (measured:
// 4.6 ms on the headline shape, every phase latency-bound).  The code is canonical -- per code
// length ascending symbols, the longest codes lowest (HUF_readDTableX1's fill order) -- so a
// block's table follows from the symbols sorted by (weight, symbol) and start[w] = the first
// table index of weight w: a look-ahead value idx of weight w has the symbol
// sorted[symoff[w] + ((idx - start[w]) >> (w - 1))] and is tableLog + 1 - w bits long.  That
// 320-byte DESCRIPTOR per chunk is what the tree passes leave in global memory and what the
// stream kernels build their LDS tables from:
//   follow -> tree<1> (segment leaders) -> copy (followers) -> tree<2> (everybody else) -> share
//   (small batches: one kernel, a wave per segment, for follow + leader + copy + share)
//   -> stream<true> (segments with one tree: one full table a wave) -> stream<false> (the rest:
//   an 8-bit prefix table per chunk).  DESIGN.md 4.4b has the measurements behind each step.
#include "../../include/sprintz_mi355x.h"

#include <hip/hip_runtime.h>

#include "lds_attr.h"

#include <cstdlib>
#include <string>
#include <atomic>
#include <type_traits>

namespace sprintz { int set_error(int code, const char* what); }   // api.hip: the library's one error sink

namespace sprintz {
// chunks from which the one-table stream kernel runs as workgroups of HUF0_BIG_WG (built: 2) waves with 2^HUF0_BIG_PLOG (built: 64) byte pieces (SPRINTZ_OPT_HUF0_BIG_BATCH)
std::atomic<long long>& huf0_big_batch()
{
    // one more than 16 chunks x the chip's 1 024 SIMDs: while every wave of the single-wave form has a SIMD to itself that form is the faster one
    // (tools/huf0_threshold.py: 16 384 chunks 127 vs 136 us, 17 000 chunks 188 vs 137)
    static std::atomic<long long> v{16385};
    return v;
}
// chunks up to which the stream stage runs as one wave per chunk with sixteen self-synchronising decoders per stream (huf0_sync.h; SPRINTZ_OPT_HUF0_SYNC_CHUNKS; 0 = never)
std::atomic<long long>& huf0_sync_chunks()
{
    static std::atomic<long long> v{[] { const char* e = getenv("SPRINTZ_MI355X_HUF0_SYNC_CHUNKS"); return e ? (atoll(e) < 0 ? 0ll : atoll(e)) : 8192ll; }()};
    return v;
}
}  // namespace sprintz

namespace {

constexpr int kWStride = 128 + 4;                // weights per chunk, a nibble each (symbol 2b = high nibble of byte b), padded off the bank stride
constexpr int kRStride = 344 + 4;                // per chunk: 8 zero bytes + header copy 144 | norm 32 (later the weight counts) | next 32 | fse 128;
                                                 // after read_stats2: the sorted symbols (256) | the table words (68)
constexpr int kCntOff = 152, kTabOff = 256;
constexpr int64_t kCorrupt = SPRINTZ_E_CORRUPT;

typedef uint32_t v2u __attribute__((ext_vector_type(2)));
typedef v2u __attribute__((aligned(1), may_alias)) v2u_a1;
typedef uint32_t __attribute__((aligned(1), may_alias)) u32_a1;

__device__ __forceinline__ int highbit(uint32_t v) { return 31 - __clz((int)v); }      // v != 0

// bytes b .. b+4 of a zero-padded LDS byte array, little endian
__device__ __forceinline__ uint64_t rd40(const uint8_t* h, uint32_t b)
{
    return (uint64_t)h[b] | ((uint64_t)h[b + 1] << 8) | ((uint64_t)h[b + 2] << 16) | ((uint64_t)h[b + 3] << 24) | ((uint64_t)h[b + 4] << 32);
}
// backward reader (BIT_DStream_t as a cursor P = unread bits): the nb (<= 16) bits below P, MSB first, 0 before the start
__device__ __forceinline__ uint32_t back_look(const uint8_t* h, int P, int nb)
{
    if (P <= 0 || nb == 0) return 0;
    const int lo = P - nb;
    if (lo >= 0) return (uint32_t)(rd40(h, (uint32_t)lo >> 3) >> (lo & 7)) & ((1u << nb) - 1u);
    return ((uint32_t)rd40(h, 0) & ((1u << P) - 1u)) << (-lo);
}

// ---- HUF_readStats (entropy_common.c) as the tree kernel runs it: one lane per chunk, every lane of the wave busy, so
// what counts is the length of the DEPENDENT chain of LDS round trips and the LDS a lane holds (it sets the waves a CU
// keeps).  Every rejection of the CPU function is kept.  The header copy `hb` carries 8 zero bytes in front and zero
// padding behind, so bit fields are two aligned dwords + v_alignbit instead of five byte reads; the two FSE states' table
// reads are issued together; the weights are kept a nibble each (wq: symbol 2b = high nibble of byte b -- the order of
// the 4-bit description, which is then just copied), written and read back eight at a time; the per-weight counts are
// LDS adds with no return (cnt[13], over the dead norm[]; on return cnt[w] = number of symbols of weight w, the implied
// last one included).  s: int16 norm[16] | u16 next[16] | u16 fse[64]: weights are < 12, so a description that gives
// probability to a symbol >= 16 is damaged.  Returns header bytes, 0 if damaged.
__device__ __forceinline__ uint32_t hb32(const uint8_t* hb, uint32_t bitpos)                       // 32 bits at bit `bitpos` of hb, LSB first
{
    const uint32_t* const q = (const uint32_t*)hb + (bitpos >> 5);
    return __builtin_amdgcn_alignbit(q[1], q[0], bitpos & 31u);
}
__device__ __forceinline__ uint32_t nib_shift(uint32_t i) { return 8u * ((i & 7u) >> 1) + ((i & 1u) ? 0u : 4u); }   // weight i in its dword
// first half: the weights of all symbols but the last (osize of them) -> wq; returns the header bytes, 0 if damaged
#ifdef HUF0_TREE_TIMING                         // experiment builds: where a leader's wave spends its time (block 0, lane 0; 10 ns ticks)
__device__ uint64_t g_tree_ts[16];
#define TREE_TS(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_tree_ts[k] = wall_clock64(); } while (0)
#else
#define TREE_TS(k) do {} while (0)
#endif

// FSE_readNCount (entropy_common.c) on the header copy: norm[0..15] (zeroed by the caller), table log, last symbol, bytes read
// UNI: every lane of the wave runs it on the same bytes: the description's first 160 bits are fetched once into wave-uniform registers
// and the walk shifts a 64-bit window along them on the scalar unit (a lone wave pays ~8 cycles an instruction and ~70 for an LDS
// round trip: the per-field LDS fetch of the lane form made this 5.8 us of the leader's 25)
template <bool UNI = false>
__device__ __forceinline__ bool fse_read_ncount(const uint8_t* hb0, uint32_t isize, int16_t* norm, uint32_t& tl_out, uint32_t& max_sv_out, uint32_t& hl_out)
{
    // FSE_readNCount: bit 0 of the description is bit 72 of hb (8 lead bytes + the size byte)
    constexpr uint32_t F0 = 72;
    uint32_t bp = 0;
    auto uni = [](uint32_t v) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    uint32_t d2 = 0, d3 = 0, d4 = 0, have = 64, idx = 2;
    uint64_t w = 0;
    if constexpr (UNI) {
        const uint32_t d0 = uni(hb32(hb0, F0)), d1 = uni(hb32(hb0, F0 + 32u));
        d2 = uni(hb32(hb0, F0 + 64u)); d3 = uni(hb32(hb0, F0 + 96u)); d4 = uni(hb32(hb0, F0 + 128u));
        w = ((uint64_t)d1 << 32) | d0;
    }
    auto peek = [&]() -> uint32_t {                       // the next 32 bits (UNI: at least 32 of the window's bits are valid)
        if constexpr (UNI) return (uint32_t)w; else return hb32(hb0, F0 + bp);
    };
    auto skip = [&](uint32_t k) {
        bp += k;
        if constexpr (UNI) {
            w >>= k;
            have -= k;
            if (have <= 32u) {
                const uint32_t nx = idx == 2u ? d2 : idx == 3u ? d3 : idx == 4u ? d4 : uni(hb32(hb0, F0 + 32u * idx));
                w |= (uint64_t)nx << have;
                have += 32u;
                idx++;
            }
        }
    };
    int nb = (int)(peek() & 0xf) + 5;
    if (nb > 6) return false;                                     // tableLog > maxLog (6)
    skip(4);
    tl_out = (uint32_t)nb;
    int remaining = (1 << nb) + 1, threshold = 1 << nb;
    nb++;
    uint32_t charnum = 0;
    bool previous0 = false;
    const uint32_t bit_end = 8u * isize;
    while (remaining > 1 && charnum <= 255u) {
        if (previous0) {
            uint32_t n0 = charnum;
            while ((peek() & 0xffffu) == 0xffffu) { n0 += 24; skip(16); if (bp > bit_end) return false; }
            while ((peek() & 3u) == 3u) { n0 += 3; skip(2); if (bp > bit_end) return false; }
            n0 += peek() & 3u;
            skip(2);
            if (n0 > 255u) return false;
            charnum = n0;                                     // norm is zero there already
        }
        const uint32_t bits = peek();
        const int max = (2 * threshold - 1) - remaining;
        int count;
        if ((int)(bits & (uint32_t)(threshold - 1)) < max) {
            count = (int)(bits & (uint32_t)(threshold - 1));
            skip((uint32_t)(nb - 1));
        } else {
            count = (int)(bits & (uint32_t)(2 * threshold - 1));
            if (count >= threshold) count -= max;
            skip((uint32_t)nb);
        }
        count--;
        remaining -= count < 0 ? -count : count;
        if (charnum > 255u || bp > bit_end) return false;
        if (charnum >= 16u) { if (count != 0) return false; charnum++; }
        else norm[charnum++] = (int16_t)count;
        previous0 = count == 0;
        while (remaining < threshold) { nb--; threshold >>= 1; }
    }
    if (remaining != 1 || bp > bit_end || charnum == 0) return false;
    max_sv_out = (charnum < 16u ? charnum : 16u) - 1;
    hl_out = (bp + 7) >> 3;
    return hl_out < isize;
}

__device__ uint32_t read_weights(const uint8_t* hb, uint32_t n, uint32_t* wq, uint8_t* s, uint32_t& osize_out)
{
    const uint8_t* const h = hb + 8;
    if (n < 1) return 0;
    uint32_t isize = h[0], osize;
    if (isize >= 128) {                                           // 4-bit weights
        osize = isize - 127;
        isize = (osize + 1) / 2;
        if (isize + 1 > n) return 0;
        for (uint32_t k = 0; 8 * k < osize; k++) wq[k] = hb32(hb, 72u + 32u * k);      // the description IS the nibbles, in this order
    } else {                                                      // FSE_decompress_wksp, table log <= 6
        if (isize + 1 > n) return 0;
        int16_t* const norm = (int16_t*)s;
        uint16_t* const next = (uint16_t*)(s + 32);
        uint16_t* const fse = (uint16_t*)(s + 64);                // symbol | nbits << 4 | new_state << 8
        for (int k = 0; k < 8; k++) ((uint32_t*)norm)[k] = 0;
        uint32_t tl = 0, max_sv = 0, hl = 0;
        if (!fse_read_ncount(hb, isize, norm, tl, max_sv, hl)) return 0;
        TREE_TS(2);
        // FSE_buildDTable
        const uint32_t size = 1u << tl;
        uint32_t high = size - 1;
        for (uint32_t sy = 0; sy <= max_sv; sy++) {
            if (norm[sy] == -1) { fse[high--] = (uint16_t)sy; next[sy] = 1; }
            else next[sy] = (uint16_t)norm[sy];
        }
        {
            const uint32_t mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
            uint32_t pos = 0, placed = 0;
            for (uint32_t sy = 0; sy <= max_sv; sy++)
                for (int i = 0; i < norm[sy]; i++) {
                    if (++placed > size) return 0;
                    fse[pos] = (uint16_t)sy;
                    pos = (pos + step) & mask;
                    while (pos > high) pos = (pos + step) & mask;
                }
            if (pos != 0) return 0;
        }
        for (uint32_t u = 0; u < size; u++) {
            const uint32_t sy = fse[u] & 0xfu, ns = next[sy]++;
            if (ns == 0 || ns >= 2 * size) return 0;
            const uint32_t nbits = tl - (uint32_t)highbit(ns);
            fse[u] = (uint16_t)(sy | (nbits << 4) | (((ns << nbits) - size) << 8));
        }
        TREE_TS(3);
        // FSE_decompress_usingDTable: two interleaved states; the stream ends by running dry
        const uint8_t* const b = h + 1 + hl;
        const uint32_t bn = isize - hl;
        if (b[bn - 1] == 0) return 0;
        int P = 8 * (int)(bn - 1) + highbit(b[bn - 1]);
        uint32_t s1 = back_look(b, P, (int)tl); P -= (int)tl;
        uint32_t s2 = back_look(b, P, (int)tl); P -= (int)tl;
        osize = 0;
        const uint32_t B0 = 8u + 1u + hl;                         // byte offset of the bit stream in hb
        // the stream's next 64 bits ride in a register, refilled every 8 weights (8 x 6 bits <= the 57 a refill guarantees)
        bool done = false;
        while (!done) {                                           // eight weights a trip: one dword of nibbles
            uint64_t win = 0;
            if (P > 0) {
                const uint32_t a = B0 + ((uint32_t)(P - 1) >> 3) - 7u;            // first of the 8 bytes that end at the cursor's byte (>= 2)
                const uint32_t* const q = (const uint32_t*)hb + (a >> 2);
                const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
                const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, a & 3u), hi = __builtin_amdgcn_alignbyte(d2, d1, a & 3u);
                win = (((uint64_t)hi << 32) | lo) << (7 - ((P - 1) & 7));
                if (P < 64) win &= ~0ull << (64 - P);             // nothing before the stream's first bit
            }
            uint32_t pack = 0, spill = 0;
            const uint32_t q0 = osize >> 3;                       // osize is a multiple of 8 here
            bool over = false;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t e1 = fse[s1], e2 = fse[s2];       // both table reads in flight together
                if (osize + 2 > 255u) return 0;
                pack |= (e1 & 0xfu) << (8 * r + 4);
                osize++;
                uint32_t nbt = (e1 >> 4) & 0xfu;
                s1 = (e1 >> 8) + ((((uint32_t)(win >> 32)) >> 1) >> (31u - nbt));
                win <<= nbt;
                P -= (int)nbt;
                if (P < 0) { pack |= (e2 & 0xfu) << (8 * r); osize++; done = true; break; }
                if (osize + 2 > 255u) return 0;
                pack |= (e2 & 0xfu) << (8 * r);
                osize++;
                nbt = (e2 >> 4) & 0xfu;
                s2 = (e2 >> 8) + ((((uint32_t)(win >> 32)) >> 1) >> (31u - nbt));
                win <<= nbt;
                P -= (int)nbt;
                if (P < 0) {
                    const uint32_t w = fse[s1] & 0xfu;
                    if (r < 3) pack |= w << (8 * (r + 1) + 4); else { spill = w << 4; over = true; }
                    osize++;
                    done = true;
                    break;
                }
            }
            wq[q0] = pack;
            if (over) wq[q0 + 1] = spill;
        }
    }
    TREE_TS(4);
    osize_out = osize;
    return isize + 1;
}

// both halves, one lane: the weights, then their statistics (eight weights a read; the last symbol's weight is implied;
// the counts take the place of norm)
__device__ uint32_t read_stats2(const uint8_t* hb, uint32_t n, uint32_t* wq, uint8_t* s, uint32_t* cnt, uint32_t& nsym, uint32_t& tl_out)
{
    uint32_t osize = 0;
    const uint32_t hlen = read_weights(hb, n, wq, s, osize);
    if (hlen == 0) return 0;
    for (int k = 0; k < 13; k++) cnt[k] = 0;
    uint32_t total = 0, rank1 = 0;
    bool heavy = false;
    for (uint32_t k = 0; k < osize; k += 8) {
        const uint32_t eight = wq[k >> 3];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t w = (eight >> (8 * (i >> 1) + ((i & 1) ? 0 : 4))) & 0xfu;
            if (k + i < osize) {
                heavy |= w >= 12u;
                rank1 += w == 1u;
                total += (1u << w) >> 1;
                __hip_atomic_fetch_add(&cnt[w < 12u ? w : 0u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        }
    }
    if (heavy) return 0;
    if (total == 0) return 0;
    const uint32_t tl = (uint32_t)highbit(total) + 1u;
    if (tl > 12u) return 0;
    const uint32_t rest = (1u << tl) - total;
    if ((1u << highbit(rest)) != rest) return 0;
    const uint32_t lw = (uint32_t)highbit(rest) + 1u;
    {
        const uint32_t sh = nib_shift(osize), old = (osize & 7u) ? wq[osize >> 3] : 0u;     // (a fresh dword if osize is a multiple of 8)
        wq[osize >> 3] = (old & ~(0xfu << sh)) | (lw << sh);
    }
    __hip_atomic_fetch_add(&cnt[lw], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    rank1 += lw == 1u;
    if (rank1 < 2 || (rank1 & 1u)) return 0;
    nsym = osize + 1;
    tl_out = tl;
    return hlen;
}

__device__ __forceinline__ int quad_bcast0(int v) { return __builtin_amdgcn_mov_dpp(v, 0x00, 0xf, 0xf, true); }
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- the same first half for huf0_tree_wave_kernel: the WAVE reads one description.  FSE_readNCount and the two-state weight
// decode stay serial (every lane runs them on the same LDS bytes: uniform control flow, nothing to broadcast); FSE_buildDTable
// -- 12 of the lane version's 49 us -- is spread over the lanes: lane k holds step k of the symbol spread (cell (k * step) &
// mask, skipped if above `high`; its rank among the cells not skipped says which symbol lands there), then lane u holds cell u
// (its `next` value = norm[symbol] + the cells of the same symbol below it, one ballot per symbol).  The weight decode runs
// without a branch inside a trip of eight weights: the step after which the cursor goes negative is looked up afterwards
// (the symbol emitted after the stream runs dry is the next step's symbol anyway -- it depends on the state alone).
// Same results and the same rejections as read_weights.  Call with all 64 lanes; t = lane.
__device__ uint32_t read_weights_wave(const uint8_t* hb, uint32_t n, uint32_t* wq, uint8_t* s, uint32_t& osize_out, const int t)
{
    const uint8_t* const h = hb + 8;
    if (n < 1) return 0;
    uint32_t isize = h[0], osize;
    auto below = [&](uint64_t m) -> uint32_t { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); };
    if (isize >= 128) {                                           // 4-bit weights
        osize = isize - 127;
        isize = (osize + 1) / 2;
        if (isize + 1 > n) return 0;
        if (8u * (uint32_t)t < osize) wq[t] = hb32(hb, 72u + 32u * (uint32_t)t);
        osize_out = osize;
        return isize + 1;
    }
    if (isize + 1 > n) return 0;
    int16_t* const norm = (int16_t*)s;
    uint16_t* const fse = (uint16_t*)(s + 64);                    // symbol | nbits << 4 | new_state << 8
    if (t < 8) ((uint32_t*)norm)[t] = 0;
    wave_sync();
    uint32_t tl = 0, max_sv = 0, hl = 0;
    if (!fse_read_ncount<true>(hb, isize, norm, tl, max_sv, hl)) return 0;
    wave_sync();
    TREE_TS(2);
    const uint32_t size = 1u << tl, mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
    {
        // cum[sy] = cells of the symbols below sy that go through the spread; the symbols of probability "-1" take the top cells
        uint32_t cum[17], nlp = 0, lp_below = 0;
        cum[0] = 0;
#pragma unroll
        for (int sy = 0; sy < 16; sy++) {
            const int c = (uint32_t)sy <= max_sv ? (int)norm[sy] : 0;
            lp_below += (c == -1 && sy < t) ? 1u : 0u;
            nlp += c == -1 ? 1u : 0u;
            cum[sy + 1] = cum[sy] + (c > 0 ? (uint32_t)c : 0u);
        }
        if (cum[16] + nlp != size) return 0;                      // (FSE_buildDTable's "placed > size" / "pos != 0")
        const uint32_t high = size - 1 - nlp;
        if (t < 16 && (uint32_t)t <= max_sv && norm[t] == -1) fse[size - 1 - lp_below] = (uint16_t)t;
        const uint32_t pk = ((uint32_t)t * step) & mask;
        const bool valid = (uint32_t)t < size && pk <= high;
        const uint32_t r = below(__ballot(valid));
        if (valid) {
            uint32_t sym = 0;
#pragma unroll
            for (int sy = 1; sy < 16; sy++) sym += r >= cum[sy] ? 1u : 0u;
            fse[pk] = (uint16_t)sym;
        }
    }
    wave_sync();
    {
        const bool in = (uint32_t)t < size;
        const uint32_t sy = in ? (uint32_t)fse[t] & 0xfu : 16u;
        uint32_t same_below = 0;
#pragma unroll
        for (int sv = 0; sv < 16; sv++) {
            const uint64_t m = __ballot(sy == (uint32_t)sv);
            same_below = sy == (uint32_t)sv ? below(m) : same_below;
        }
        bool bad = false;
        uint32_t entry = 0;
        if (in) {
            const int c = (int)norm[sy];
            const uint32_t ns = (c == -1 ? 1u : (uint32_t)c) + same_below;
            bad = ns == 0 || ns >= 2 * size;
            const uint32_t nbits = tl - (uint32_t)highbit(ns | 1u);
            entry = sy | (nbits << 4) | (((ns << nbits) - size) << 8);
        }
        if (__ballot(bad)) return 0;
        wave_sync();                                              // every lane has read its cell's symbol
        if (in) fse[t] = (uint16_t)entry;
    }
    wave_sync();
    TREE_TS(3);
    // FSE_decompress_usingDTable: two interleaved states; the stream ends by running dry.  The table rides in a register (lane u
    // = cell u) and is read with v_readlane; states, window and cursor are wave-uniform, so the chain runs on the scalar unit
    const auto uni = [](uint32_t v) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    // (in the register: new_state << 8 | nbits << 4, the symbol in the TOP nibble -- a 64-bit shift of {weights so far : entry} by 4 then appends it)
    const uint32_t cell = (uint32_t)t < size ? (uint32_t)fse[t] : 0u;
    const uint32_t tabv = (cell & 0xfff0u) | (cell << 28);
    const uint8_t* const b = h + 1 + hl;
    const uint32_t bn = isize - hl;
    const uint32_t lastb = uni(b[bn - 1]);
    if (lastb == 0) return 0;
    int P = 8 * (int)(bn - 1) + highbit(lastb);
    uint32_t s1 = uni(back_look(b, P, (int)tl)); P -= (int)tl;
    uint32_t s2 = uni(back_look(b, P, (int)tl)); P -= (int)tl;
    const uint32_t B0 = 8u + 1u + hl;                             // byte offset of the bit stream in hb
    int endk = -1;                                                // the step (= weight index) after which the cursor is negative
    auto window_at = [&](int Pc) -> uint64_t {                    // the stream's next 64 bits below cursor Pc (8 x 6 bits <= the 57 a refill guarantees)
        uint64_t win = 0;
        if (Pc > 0) {
            const uint32_t a = B0 + ((uint32_t)(Pc - 1) >> 3) - 7u;
            const uint32_t* const q = (const uint32_t*)hb + (a >> 2);
            const uint32_t d0 = uni(q[0]), d1 = uni(q[1]), d2 = uni(q[2]);
            const uint64_t lo = ((((uint64_t)d1 << 32) | d0) >> (8u * (a & 3u))) & 0xffffffffull, hi = ((((uint64_t)d2 << 32) | d1) >> (8u * (a & 3u))) & 0xffffffffull;
            win = ((hi << 32) | lo) << (7 - ((Pc - 1) & 7));
            if (Pc < 64) win &= ~0ull << (64 - Pc);               // nothing before the stream's first bit
        }
        return win;
    };
    auto take = [&](uint32_t e, uint32_t& st, uint64_t& win, int& Pc) {      // one state's step: ten scalar instructions with the append below
        const uint32_t nbt = (e >> 4) & 0xfu;
        st = ((e >> 8) & 0xffu) + (uint32_t)(((win >> 32) << nbt) >> 32);
        win <<= nbt;
        Pc -= (int)nbt;
    };
    for (uint32_t trip = 0; trip < 32u; trip++) {                 // eight weights a trip: one dword of nibbles
        const int P0 = P;
        const uint32_t s10 = s1, s20 = s2;
        uint64_t win = window_at(P);
        uint32_t pack = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t e1 = (uint32_t)__builtin_amdgcn_readlane((int)tabv, (int)s1), e2 = (uint32_t)__builtin_amdgcn_readlane((int)tabv, (int)s2);
            pack = (uint32_t)(((((uint64_t)pack << 32) | e1) << 4) >> 32);
            take(e1, s1, win, P);
            pack = (uint32_t)(((((uint64_t)pack << 32) | e2) << 4) >> 32);
            take(e2, s2, win, P);
        }
        pack = __builtin_bswap32(pack);                           // weight i of the trip: byte i / 2, an even i in the high nibble
        if (P < 0) {                                              // the stream ran dry inside this trip: once more, step by step, for where
            int Pr = P0;
            uint32_t r1 = s10, r2 = s20, j = 8;
            uint64_t wr = window_at(P0);
            for (uint32_t k = 0; k < 8u; k++) {
                uint32_t& st = (k & 1u) ? r2 : r1;
                take((uint32_t)__builtin_amdgcn_readlane((int)tabv, (int)st), st, wr, Pr);
                if (Pr < 0) { j = k; break; }
            }
            // steps 0 .. j of this trip are the stream's, step j + 1 is the symbol emitted after it ran dry
            endk = (int)(8u * trip + j);
            if (j == 7u) {
                wq[trip] = pack;
                wq[trip + 1] = ((uint32_t)__builtin_amdgcn_readlane((int)tabv, (int)s1) >> 28) << 4;
            } else {
                const uint32_t m = j + 1u;                        // weights 0 .. m of the dword stay
                const uint32_t keep = (m & 1u) ? ((m >> 1) == 3u ? 0xffffffffu : (1u << (8u * ((m >> 1) + 1u))) - 1u)
                                               : ((1u << (8u * (m >> 1))) - 1u) | (0xf0u << (8u * (m >> 1)));
                wq[trip] = pack & keep;
            }
            break;
        }
        wq[trip] = pack;
    }
    if (endk < 0 || endk > 253) return 0;                          // (read_weights: "osize + 2 > 255" before a weight is taken)
    TREE_TS(4);
    osize_out = (uint32_t)endk + 2u;
    return isize + 1;
}

// ---------------------------------------------------------------------------------------------
// The tree passes.  huf0_tree_kernel: ONE LANE PER CHUNK (64 chunks a wave) turns a coded block's tree
// description into the descriptor: the symbols sorted by (weight, symbol), start[w] | symoff[w] << 16 for
// w = 1..12, and the header length / table log.  (Round 1 ran this on the first lane of each quad of the
// stream decoder -- a quarter of the lanes for a third of its instructions, 0.5 of 1.4 ms on the headline shape.)
// The stream kernels: lane = (chunk, stream), 16 chunks a wave.  Per-chunk tables (SO = false): the top 8
// bits of the look-ahead index a 256-entry table in LDS that resolves every code of <= 8 bits
// (symbol | length << 8); longer codes can only have the weights 1 .. tableLog - 8 <= 4, so THEIR weight is
// three compares against start[2..4] -- or sits in the entry when the prefix holds one weight only -- and
// their symbol one more LDS read from the sorted list.  One table for the wave (SO = true): see below.
constexpr uint32_t kSharedMaxLog = 11;           // table log up to which a segment of one tree decodes through ONE full table
constexpr int kDescStride = 320;                 // sorted[256] | u32 tab[16]: [0] = hl | tl << 16, [w] = start[w] | symoff[w] << 16
constexpr int kCStride = 256 + 64 + 512 + 4;     // stream kernel, per chunk in LDS: sorted | tab | table8; odd in dwords
// stream kernel, per lane: a ring of two PIECES of its stream (+ the first 8 bytes again).  A piece is 16 bytes for the
// general kernel and 64 for the one-table kernel, which has the LDS to spare (see the note on memory traffic there).
// (Measured at 800 000 chunks, one-table kernel: 64-byte pieces +13 %, a workgroup of 4 waves around one table +3 %,
// both together +3 % against 16-byte pieces and a table per wave: what they save in requests they lose in resident waves.)
// (Round 3, with the streams' last bursts in one common round -- see the stream kernel: at 800 000 chunks 32-byte pieces and a
// workgroup of 4 waves around one table take the stage from 3.44 to 3.05 - 3.13 ms (FETCH_SIZE says why: half the useless lines);
// 16-byte pieces with the 4-wave workgroup 3.69, 64-byte pieces 4.13 / 3.59 (1 / 4 waves).  At 10 000 chunks the single-wave,
// 16-byte form is 5 % faster -- a workgroup there is a barrier and a four times longer table build for nothing -- so both are
// instantiated and the launch picks by batch size.)
#ifndef HUF0_SPECULATIVE_TAIL
#define HUF0_SPECULATIVE_TAIL 1           // the streams' last bursts in one common masked round (stream kernel)
#endif
constexpr int ring_stride(int plog) { return 2 * (1 << plog) + 8; }
#ifndef HUF0_G_PLOG
#define HUF0_G_PLOG 5                     // the per-chunk-table kernel: piece size, and whether it refills on the cadence
#endif
#ifndef HUF0_G_CAD
#define HUF0_G_CAD 1
#endif
#ifndef HUF0_SMALL_PLOG
#define HUF0_SMALL_PLOG 6                 // the single-wave one-table kernel of small batches: piece size, and whether it refills on the cadence
#endif
#ifndef HUF0_SMALL_CAD
#define HUF0_SMALL_CAD 1
#endif
#ifndef HUF0_BIG_NS
#define HUF0_BIG_NS 3                     // ring slots of the big-batch one-table kernel (2: 64-byte pieces only)
#endif
#ifndef HUF0_FAST_TAILS
#define HUF0_FAST_TAILS 1                 // every round that is not a fast one (a wave's first, cut at the output's lines, and its last) runs its whole steps as fast steps + ONE masked step (round 5; before: the unaligned-burst forms' last round only)
#endif
#ifndef HUF0_BIG_QW
#define HUF0_BIG_QW 1                     // the big-batch one-table kernel: a quad fetches each of its four streams' pieces together (one 64-byte request instead of four of 16 bytes) and its bursts leave as whole 128-byte lines (round 5; tools/probes/huf0_pattern.hip)
#endif
#ifndef HUF0_QW_PAIR
#define HUF0_QW_PAIR 0                    // the quad-wide form pairs its 64-byte bursts into whole 128-byte lines
#endif
#ifndef HUF0_QW_WAVES
#define HUF0_QW_WAVES 3                   // register budget of the quad-wide form as waves a SIMD (3: 168 VGPRs, 18 spilled; 2: 256)
#endif
#ifndef HUF0_WAVE_LEADERS
#define HUF0_WAVE_LEADERS 65536          // segments up to which the leaders' trees are parsed a WAVE each (one kernel that also does the follow test, the copies and the share flags); above: a lane each + four small passes
#endif
#ifndef HUF0_BIG_UA
#define HUF0_BIG_UA 0                     // the big-batch one-table kernel's bursts start where a stream's output starts (no masked first round) instead of on 64-byte lines
#endif
#ifndef HUF0_BIG_WG
#define HUF0_BIG_WG 2                     // wavefronts a workgroup of the big-batch one-table kernel
#endif
#ifndef HUF0_BIG_PLOG
#define HUF0_BIG_PLOG 6                   // log2 of the stream piece of the big-batch one-table kernel
#endif
#ifndef HUF0_CARRY_WINDOW
#define HUF0_CARRY_WINDOW 1              // the one-table path carries its 64-bit window from step to step (fast_step; round 5)
#endif
#ifndef HUF0_CARRY_RINGPOS
#define HUF0_CARRY_RINGPOS 1             // ... and where the bits below it sit in the three-slot ring (a bit position modulo 8 * 192)
#endif
#ifndef HUF0_CADENCED
#define HUF0_CADENCED 1                  // the 4-wave one-table kernel refills on a fixed cadence (template parameter CAD)
#endif

// Blocks written with one code per SEGMENT of 64 chunks (our writer; any writer that repeats a tree description) need
// the tree only once per segment.  follow[c] = 1 iff chunk c is not the first of its 64-aligned segment and its tree
// description is byte-for-byte the one of the segment's first chunk: such a chunk takes a copy of that chunk's
// descriptor (huf0_copy_kernel) instead of parsing.  The tree kernel then runs twice: over the segment leaders (a lane
// per leader: 4096 chunks a wave), and over whatever is left (libzstd's blocks: everything), waves with nothing left
// exiting at once.  Per wave the tree is ~115 us of serial work; this takes it from ceil(chunks / 64 / 768) rounds to one.
__device__ __forceinline__ uint32_t tree_desc_bytes(uint32_t b0) { return b0 < 128u ? 1u + b0 : 1u + (b0 - 127u + 1u) / 2u; }   // HUF_readStats: iSize + 1

__device__ __forceinline__ uint8_t follows_leader(const uint8_t* __restrict__ blocks, const uint64_t* __restrict__ boffs,
                                                  const uint64_t* __restrict__ ooffs, uint64_t c)
{
    const uint64_t L = c & ~(uint64_t)63;
    uint8_t f = 0;
    if (c != L) {
        const uint64_t b0 = boffs[c], b1 = boffs[c + 1], l0 = boffs[L], l1 = boffs[L + 1];
        const uint64_t cs = b1 - b0, ls = l1 - l0, cd = ooffs[c + 1] - ooffs[c], ld = ooffs[L + 1] - ooffs[L];
        if (b1 >= b0 && l1 >= l0 && cs > 1 && ls > 1 && cs < cd && ls < ld) {      // both are coded blocks
            const uint8_t* const p = blocks + b0;
            const uint8_t* const q = blocks + l0;
            const uint32_t hl = tree_desc_bytes(p[0]);
            if (hl < cs && hl < ls) {
                typedef uint64_t __attribute__((aligned(1))) u64_a1;
                uint64_t diff = 0;                                // no early exit: the loads of one lane are all in flight together
                if (hl >= 8) {
                    for (uint32_t k = 0; k + 8 <= hl; k += 8) diff |= *(const u64_a1*)(p + k) ^ *(const u64_a1*)(q + k);
                    diff |= *(const u64_a1*)(p + hl - 8) ^ *(const u64_a1*)(q + hl - 8);       // the tail, overlapping
                } else {
                    for (uint32_t k = 0; k < hl; k++) diff |= (uint64_t)(p[k] ^ q[k]);
                }
                f = diff == 0 ? 1 : 0;
            }
        }
    }
    return f;
}

__global__ void __launch_bounds__(256) huf0_follow_kernel(const uint8_t* __restrict__ blocks, const uint64_t* __restrict__ boffs,
                                                          const uint64_t* __restrict__ ooffs, uint64_t nchunks, uint8_t* __restrict__ follow)
{
    const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (c < nchunks) follow[c] = follows_leader(blocks, boffs, ooffs, c);
}

// thread = one 16-byte piece of one follower's descriptor
// (a segment that share[] gives to the one-table kernel needs no copies: that kernel reads the LEADER's descriptor -- 256 MB of descriptors
//  not written at 800 000 chunks)
__global__ void __launch_bounds__(256) huf0_copy_kernel(uint8_t* __restrict__ desc, const uint8_t* __restrict__ follow, uint64_t nchunks,
                                                        const uint8_t* __restrict__ share)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x, c = i / 20, part = i % 20;
    if (c >= nchunks || !follow[c] || share[c >> 6]) return;
    const uint64_t L = c & ~(uint64_t)63;
    *(uint4*)(desc + c * kDescStride + 16 * part) = *(const uint4*)(desc + L * kDescStride + 16 * part);
}

// MODE 1: lane t of workgroup g parses the segment leader 64 (64 g + t).  MODE 2: lane t parses chunk 64 g + t unless it is a
// leader or a follower.
template <int MODE>
__device__ __forceinline__ void huf0_tree_body(const uint8_t* __restrict__ blocks, const uint64_t* __restrict__ boffs,
                                               const uint64_t* __restrict__ ooffs, uint64_t nchunks, uint8_t* __restrict__ desc,
                                               const uint8_t* __restrict__ follow, const uint64_t group)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_w[64 * kWStride];
    __shared__ __attribute__((aligned(16))) uint8_t s_r[64 * kRStride];
    __shared__ uint64_t s_src[64];
    __shared__ uint32_t s_hc[64];
    __shared__ uint64_t s_chunk[64];                              // the chunk a lane works on, or ~0: nothing to write back
    const int t = threadIdx.x;
    const uint64_t lane_item = group * 64 + (uint64_t)t;
    const uint64_t chunk = MODE == 1 ? lane_item * 64 : lane_item;
    bool exists = chunk < nchunks;
    if (MODE == 2 && exists) exists = (chunk & 63) != 0 && follow[chunk] == 0;
    if (__ballot(exists) == 0) return;                            // a wave of followers: nothing to parse
    s_chunk[t] = exists ? chunk : ~0ull;
    const uint64_t b0 = exists ? boffs[chunk] : 0, b1 = exists ? boffs[chunk + 1] : 0;
    const uint64_t o0 = exists ? ooffs[chunk] : 0, o1 = exists ? ooffs[chunk + 1] : 0;
    const uint64_t csize = b1 - b0, dsize = o1 - o0;
    const bool coded = exists && b1 >= b0 && o1 >= o0 && csize > 1 && csize < dsize;      // HUF_decompress's third case
    const uint32_t hcopy = coded ? (uint32_t)(csize < 129 ? csize : 129) : 0u;
    s_src[t] = (uint64_t)(uintptr_t)(blocks + b0);
    s_hc[t] = hcopy;
    wave_sync();
    // the wave copies one chunk's header per trip: 38 lanes, one (unaligned) dword each -- 8 zero bytes, the 129 header bytes
    // (one request), zero padding up to byte 152
    for (int c0 = 0; c0 < 64; c0 += 8) {                          // eight chunks a trip, their loads in flight together
        uint32_t v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t hc = s_hc[c0 + i];
            const uint8_t* const src = (const uint8_t*)(uintptr_t)s_src[c0 + i];
            v[i] = 0;
            if (t >= 2 && t < 38) {
                const uint32_t k = 4u * (uint32_t)(t - 2);
                if (k + 4 <= hc) v[i] = *(const u32_a1*)(src + k);
                else for (uint32_t bb = 0; bb < 4 && k + bb < hc; bb++) v[i] |= (uint32_t)src[k + bb] << (8 * bb);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; i++)
            if (t < 38) *(uint32_t*)(s_r + (c0 + i) * kRStride + 4u * (uint32_t)t) = v[i];
    }
    wave_sync();
    uint32_t* const wq = (uint32_t*)(s_w + t * kWStride);
    uint8_t* const scratch = s_r + t * kRStride;
    uint8_t* const sorted = scratch;
    uint32_t* const tab = (uint32_t*)(scratch + kTabOff);         // over the FSE table, dead by then
    uint32_t hl = 0, nsym = 0, tl = 0;
    uint32_t cnt[13];
#pragma unroll
    for (int w = 0; w < 13; w++) cnt[w] = 0;
    if (coded) {
        uint32_t* const c = (uint32_t*)(scratch + kCntOff);
        hl = read_stats2(scratch, hcopy, wq, scratch + 152, c, nsym, tl);
        if (hl >= csize) hl = 0;
        if (hl) {
#pragma unroll
            for (int w = 1; w < 13; w++) cnt[w] = c[w];
        }
    }
    // start[w] (first table index of weight w) | symoff[w] << 16; the symbols sorted by (weight, symbol)
    {
        uint32_t at = 0, so = 0;
#pragma unroll
        for (int w = 1; w < 13; w++) {
            tab[w] = at | (so << 16);
            at += cnt[w] << (w - 1);
            so += cnt[w];
        }
        tab[0] = hl | (tl << 16);
        tab[13] = tab[14] = tab[15] = tab[16] = 0;
    }
    if (hl) {
        // (zeroed first: blocks with the same tree description must give byte-identical descriptors -- the stream kernel
        //  shares one decode table among the chunks of a wave when they do)
        for (int k = 0; k < 64; k++) ((uint32_t*)sorted)[k] = 0;
        // counting sort, eight symbols a trip: their slots come back from eight LDS adds issued together (DS operations
        // of a wave execute in issue order, so equal weights keep their symbol order).  The running slot of weight w is
        // the upper half of tab[w]; the table words are written again afterwards.
        for (uint32_t sy = 0; sy < nsym; sy += 8) {
            const uint32_t eight = wq[sy >> 3];
            uint32_t pos[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint32_t w = (eight >> (8 * (i >> 1) + ((i & 1) ? 0 : 4))) & 0xfu;
                pos[i] = 0xffffffffu;
                if (sy + i < nsym && w) pos[i] = __hip_atomic_fetch_add(&tab[w], 0x10000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) >> 16;
            }
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (pos[i] != 0xffffffffu) sorted[pos[i] & 0xffu] = (uint8_t)(sy + i);
        }
        uint32_t at = 0, so = 0;
#pragma unroll
        for (int w = 1; w < 13; w++) {
            tab[w] = at | (so << 16);
            at += cnt[w] << (w - 1);
            so += cnt[w];
        }
    }
    wave_sync();
    // descriptors out: 64 dwords of sorted symbols + 13 words of table per chunk, one chunk per trip
    for (int c = 0; c < 64; c++) {
        const uint64_t ch = s_chunk[c];
        if (ch == ~0ull) continue;
        uint8_t* const d = desc + ch * kDescStride;
        const uint32_t v = *(const uint32_t*)(s_r + c * kRStride + 4 * t);
        *(uint32_t*)(d + 4 * t) = v;
        if (t < 16) *(uint32_t*)(d + 256 + 4 * t) = t < 13 ? *(const uint32_t*)(s_r + c * kRStride + kTabOff + 4 * t) : 0u;   // all of it defined: equal trees, equal bytes
    }
}

template <int MODE>
__global__ void __launch_bounds__(64) huf0_tree_kernel(const uint8_t* __restrict__ blocks, const uint64_t* __restrict__ boffs,
                                                       const uint64_t* __restrict__ ooffs, uint64_t nchunks, uint8_t* __restrict__ desc,
                                                       const uint8_t* __restrict__ follow)
{
    huf0_tree_body<MODE>(blocks, boffs, ooffs, nchunks, desc, follow, (uint64_t)blockIdx.x);
}

// One WAVE per segment leader.  A lane's tree parse is ~112 us of dependent LDS round trips however few lanes run, and the
// leaders are few (157 at BASELINE config 4's 10 000 chunks): lane 0 keeps what is serial -- FSE_readNCount, the FSE
// table, the two-state weight decode (read_weights) -- and the wave does the rest in a handful of ballots: header copy,
// weight statistics (HUF_readStats' checks, all kept), the per-weight prefix and the counting sort (a symbol's slot =
// its weight's running offset + the set lanes below it in the ballot of that weight).  Same descriptor, byte for byte.
template <bool MERGE>
__global__ void __launch_bounds__(64) huf0_tree_wave_kernel(const uint8_t* __restrict__ blocks, const uint64_t* __restrict__ boffs,
                                                            const uint64_t* __restrict__ ooffs, uint64_t nchunks, uint8_t* __restrict__ desc,
                                                            uint8_t* __restrict__ follow, uint8_t* __restrict__ share)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_hb[152 + 192 + 8];      // header copy (8 zero bytes in front) | norm, next, fse
    __shared__ __attribute__((aligned(16))) uint32_t s_wq[36];
    __shared__ __attribute__((aligned(16))) uint8_t s_sorted[256];
    __shared__ uint32_t s_cnt[16];                                            // symbols per weight
    const int t = threadIdx.x;
    const uint64_t chunk = (uint64_t)blockIdx.x * 64;
    if (chunk >= nchunks) return;
    TREE_TS(0);
    // lane t is also chunk t of the segment (does it follow the leader?): its offsets travel with the leader's, its first byte with the
    // leader's header -- the same two round trips
    const uint64_t mine_c = chunk + (uint64_t)t;
    const bool mine_exists = mine_c < nchunks;
    const uint64_t mb0 = mine_exists ? boffs[mine_c] : 0, mb1 = mine_exists ? boffs[mine_c + 1] : 0;
    const uint64_t mo0 = mine_exists ? ooffs[mine_c] : 0, mo1 = mine_exists ? ooffs[mine_c + 1] : 0;
    const uint64_t b0 = boffs[chunk], b1 = boffs[chunk + 1], o0 = ooffs[chunk], o1 = ooffs[chunk + 1];
    const uint64_t csize = b1 - b0, dsize = o1 - o0;
    const bool coded = b1 >= b0 && o1 >= o0 && csize > 1 && csize < dsize;      // HUF_decompress's third case
    const uint32_t hcopy = coded ? (uint32_t)(csize < 129 ? csize : 129) : 0u;
    const uint64_t mcs = mb1 - mb0, mcd = mo1 - mo0;
    const bool mcoded = t != 0 && mine_exists && coded && mb1 >= mb0 && mcs > 1 && mcs < mcd;      // a coded block, as the leader's is
    const uint32_t mfirst = mcoded ? (uint32_t)blocks[mb0] : 0u;
    if (t < 38) {
        uint32_t v = 0;
        if (t >= 2) {
            const uint8_t* const src = blocks + b0;
            const uint32_t k = 4u * (uint32_t)(t - 2);
            if (k + 4 <= hcopy) v = *(const u32_a1*)(src + k);
            else for (uint32_t bb = 0; bb < 4 && k + bb < hcopy; bb++) v |= (uint32_t)src[k + bb] << (8 * bb);
        }
        ((uint32_t*)s_hb)[t] = v;
    }
    if (t < 16) s_cnt[t] = 0;
    wave_sync();
    TREE_TS(1);
    // lane t is also chunk t of the segment: does it follow the leader?  (what huf0_follow_kernel, huf0_copy_kernel and
    // huf0_share_kernel do for large batches happens here: three launches less where a launch is 2 % of the job)
    uint8_t fol = 0;
    uint32_t hl = 0, osize = 0;
    // follows_leader(), in two halves: a chunk's own description bytes are REQUESTED now and looked at after the leader's parse --
    // three dependent global round trips (~4.5 us) that then pass under the parse; the leader's bytes are the header copy in LDS
    uint64_t fw[16], ftail = 0;
    uint32_t fhl = 0;
    bool fcand = false;
#pragma unroll
    for (int k = 0; k < 16; k++) fw[k] = 0;
    if (mcoded) {
        typedef uint64_t __attribute__((aligned(1))) u64_a1;
        {
            const uint8_t* const p = blocks + mb0;
            const uint32_t h = tree_desc_bytes(mfirst);
            if (h < mcs && h < csize) {
                fcand = true;
                fhl = h;
                if (h >= 8) {
#pragma unroll
                    for (int k = 0; k < 16; k++)
                        if (8u * (uint32_t)k + 8u <= h) fw[k] = *(const u64_a1*)(p + 8 * k);
                    ftail = *(const u64_a1*)(p + h - 8);          // the tail, overlapping
                } else {
                    for (uint32_t k = 0; k < h; k++) ftail |= (uint64_t)p[k] << (8u * k);
                }
            }
        }
    }
    if (coded) {                                               // (wave-uniform: the leader's sizes)
        hl = read_weights_wave(s_hb, hcopy, s_wq, s_hb + 152, osize, t);
        if (hl >= csize) hl = 0;
    }
    if (fcand) {
        const uint8_t* const lead = s_hb + 8;                  // the leader's block from its first byte (read_weights_wave leaves the copy alone)
        uint64_t diff = 0;
        if (fhl >= 8) {
#pragma unroll
            for (int k = 0; k < 16; k++)
                if (8u * (uint32_t)k + 8u <= fhl) diff |= fw[k] ^ *(const uint64_t*)(lead + 8 * k);
            diff |= ftail ^ ((uint64_t)hb32(s_hb, 8u * fhl) | ((uint64_t)hb32(s_hb, 8u * fhl + 32u) << 32));
        } else {
            uint64_t lt = 0;
            for (uint32_t k = 0; k < fhl; k++) lt |= (uint64_t)lead[k] << (8u * k);
            diff = ftail ^ lt;
        }
        fol = diff == 0 ? 1 : 0;
    }
    if (mine_exists) follow[mine_c] = fol;
    wave_sync();
    TREE_TS(5);
    hl = (uint32_t)__builtin_amdgcn_readlane((int)hl, 0);
    osize = (uint32_t)__builtin_amdgcn_readlane((int)osize, 0);
    auto weight_of = [&](uint32_t sy) -> uint32_t { return (s_wq[sy >> 3] >> nib_shift(sy)) & 0xfu; };
    auto below = [&](uint64_t m) -> uint32_t { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); };
    uint32_t tl = 0, nsym = 0;
    uint32_t cnt[13];
#pragma unroll
    for (int w = 0; w < 13; w++) cnt[w] = 0;
    if (hl) {
        // the statistics as an LDS histogram (a lane adds its four symbols' weights; 48 ballots did this before: 3.1 us of the leader's 24)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t sy = 64u * i + (uint32_t)t;
            if (sy < osize) __hip_atomic_fetch_add(&s_cnt[weight_of(sy)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        wave_sync();
        uint32_t total = 0;
#pragma unroll
        for (int ww = 0; ww < 12; ww++) {
            cnt[ww] = s_cnt[ww];
            total += ww ? cnt[ww] << (ww - 1) : 0u;
        }
        const bool heavy = (s_cnt[12] | s_cnt[13] | s_cnt[14] | s_cnt[15]) != 0u;
        bool ok = !heavy && total != 0;
        uint32_t lw = 0;
        if (ok) {
            tl = (uint32_t)highbit(total) + 1u;
            ok = tl <= 12u;
            if (ok) {
                const uint32_t rest = (1u << tl) - total;
                ok = (1u << highbit(rest)) == rest;
                lw = (uint32_t)highbit(rest) + 1u;
            }
        }
        if (ok) {
            if (t == 0) {
                const uint32_t sh = nib_shift(osize), old = (osize & 7u) ? s_wq[osize >> 3] : 0u;
                s_wq[osize >> 3] = (old & ~(0xfu << sh)) | (lw << sh);
            }
#pragma unroll
            for (int ww = 1; ww < 13; ww++) cnt[ww] += lw == (uint32_t)ww ? 1u : 0u;
            ok = cnt[1] >= 2u && (cnt[1] & 1u) == 0u;
            nsym = osize + 1;
        }
        if (!ok) { hl = 0; tl = 0; }
        wave_sync();
    }
    // start[w] (first table index of weight w) | symoff[w] << 16
    uint32_t tabw[13], run[13];
    {
        uint32_t at = 0, so = 0;
        tabw[0] = hl | (tl << 16);
        run[0] = 0;
#pragma unroll
        for (int w = 1; w < 13; w++) {
            const uint32_t c = hl ? cnt[w] : 0u;
            tabw[w] = at | (so << 16);
            run[w] = so;
            at += c << (w - 1);
            so += c;
        }
    }
    ((uint32_t*)s_sorted)[t] = 0;
    wave_sync();
    TREE_TS(6);
    if (hl) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t sy = 64u * i + (uint32_t)t;
            const uint32_t w = sy < nsym ? weight_of(sy) : 0u;
#pragma unroll
            for (int ww = 1; ww < 13; ww++) {
                const uint64_t m = __ballot(w == (uint32_t)ww);
                if (w == (uint32_t)ww) s_sorted[(run[ww] + below(m)) & 0xffu] = (uint8_t)sy;
                run[ww] += (uint32_t)__builtin_popcountll(m);
            }
        }
    }
    wave_sync();
    TREE_TS(7);
    uint8_t* const d = desc + chunk * kDescStride;
    *(uint32_t*)(d + 4 * t) = ((const uint32_t*)s_sorted)[t];
    if (t < 16) {
        uint32_t v = 0;
#pragma unroll
        for (int w = 0; w < 13; w++) v = t == w ? tabw[w] : v;
        *(uint32_t*)(d + 256 + 4 * t) = v;
    }
    // huf0_share_kernel's verdict for this segment
    const bool all_follow = __ballot(mine_exists && t != 0 && !fol) == 0;
    const bool shared_seg = all_follow && nchunks - chunk > 1 && hl != 0 && tl != 0 && tl <= kSharedMaxLog;
    // the followers' copies, a lane each: 16 pieces of sorted symbols from LDS, 4 of table words from the registers
    // (not in a segment the one-table kernels take: they read the leader's descriptor -- 320 bytes a chunk not written)
    if (fol && !shared_seg) {
        typedef uint32_t v4u __attribute__((ext_vector_type(4)));
        uint8_t* const dm = desc + mine_c * kDescStride;
#pragma unroll
        for (int k = 0; k < 16; k++) *(v4u*)(dm + 16 * k) = ((const v4u*)s_sorted)[k];
        *(v4u*)(dm + 256) = v4u{tabw[0], tabw[1], tabw[2], tabw[3]};
        *(v4u*)(dm + 272) = v4u{tabw[4], tabw[5], tabw[6], tabw[7]};
        *(v4u*)(dm + 288) = v4u{tabw[8], tabw[9], tabw[10], tabw[11]};
        *(v4u*)(dm + 304) = v4u{tabw[12], 0u, 0u, 0u};
    }
    if (t == 0) share[blockIdx.x] = shared_seg ? 1 : 0;
    TREE_TS(8);
    // chunks of this segment that are neither its leader nor followers (a writer with a tree per chunk: all of them) parse their own
    // description, a lane each -- huf0_tree_kernel<2>'s work, here instead of in a launch of its own that finds nothing to do
    // (the lanes read back the follow flags they wrote themselves)
    // (MERGE: up to 1 024 leaders -- the lane-per-chunk parse holds 32 KB of LDS a wave, which would cost larger batches resident waves)
    if constexpr (MERGE) {
        if (!all_follow) huf0_tree_body<2>(blocks, boffs, ooffs, nchunks, desc, follow, (uint64_t)blockIdx.x);
    }
}

// share[s] = 1 iff the chunks of segment s (64, fewer in the last one) all follow the segment's leader (one tree, so one descriptor)
// and its table log is at most kSharedMaxLog: the segment is the one-table kernel's.  thread = segment.
__global__ void __launch_bounds__(256) huf0_share_kernel(const uint8_t* __restrict__ desc, const uint8_t* __restrict__ follow, uint64_t nchunks,
                                                         uint8_t* __restrict__ share)
{
    const uint64_t sg = (uint64_t)blockIdx.x * 256 + threadIdx.x, c0 = sg * 64;
    if (c0 >= nchunks) return;
    bool ok = true;
    if (c0 + 64 > nchunks) {                                      // the last, short segment: the chunks that exist
        for (uint64_t c = c0 + 1; c < nchunks; c++) ok = ok && follow[c] == 1;
        const uint32_t tab0 = *(const uint32_t*)(desc + c0 * kDescStride + 256);
        ok = ok && nchunks - c0 > 1 && (tab0 & 0xffffu) != 0u && (tab0 >> 16) <= kSharedMaxLog && (tab0 >> 16) != 0u;
    } else {
        const uint4* const f = (const uint4*)(follow + c0);       // 64-byte aligned
        uint32_t all = 0x01010101u, first = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint4 v = f[k];
            if (k == 0) { first = v.x; all &= v.x | 1u; } else all &= v.x;
            all &= v.y & v.z & v.w;
        }
        const uint32_t tab0 = *(const uint32_t*)(desc + c0 * kDescStride + 256);
        ok = all == 0x01010101u && (first & 0xffu) == 0 && (tab0 & 0xffffu) != 0u && (tab0 >> 16) <= kSharedMaxLog && (tab0 >> 16) != 0u;
    }
    share[sg] = ok ? 1 : 0;
}

// Two instantiations, launched back to back; every chunk is taken by exactly one of them:
//   SO = true   workgroup = a 64-chunk segment that share[] marks: 4 waves around ONE full decode table of 2^tableLog entries
//               (4.4 KB) -- a symbol is one LDS read -- and 64-byte stream pieces;
//   SO = false  wave = 16 chunks of every other segment: sixteen 8-bit prefix tables (13 KB); a wave whose chunks do share a
//               tree (table log 12, or a short last segment) still builds the one table, over the prefix tables' space.
#ifndef HUF0_SO_WAVES
#define HUF0_SO_WAVES 5
#endif
#ifndef HUF0_G_WAVES
#define HUF0_G_WAVES 2
#endif
// WG: wavefronts of a workgroup around the one table (SO only: 1, 2 or 4 -- a workgroup stays inside one 64-chunk segment);
// PLOG: log2 of the stream piece a lane fetches at once
// CAD: the stream pieces are requested on a FIXED CADENCE instead of when a lane's cursor crosses into its next piece.  A wavefront's 64
//      lanes cross at different steps, so with the crossing-driven refill some lane parks the piece it had in flight and requests another in
//      nearly every step, inside a divergent branch: hipcc cannot count those loads and puts `s_waitcnt vmcnt(0)` before every park -- with
//      gfx950's one in-order counter the whole wave waits, every step, for the request another lane issued a step earlier and for the stores
//      of the last burst (ablations at 800 000 chunks: the loads cost 1.27 of the stage's 3.20 ms, the stores 0.8, the symbol chain 1.67).
//      Here (32-byte pieces, a ring of THREE per lane): once per group of four steps -- which take at most 22 bytes, so a cursor crosses at
//      most one piece boundary per group -- every lane parks the piece it requested one group earlier and requests the next one below what
//      it holds if the slot is free, through unconditional buffer loads (a lane that requests nothing asks for an offset outside the
//      descriptor: zeros, no traffic); the block stores are unconditional buffer stores as well.  Every VMEM operation of a round is then
//      countable and hipcc emits the `vmcnt(N)` that waits for the loads of four steps ago and nothing younger (decode_fast.h's scheme).
//      Invariant at the top of a group, after the park: pieces cursor .. cursor - 1 are in the ring (the four steps read at most 29 bytes
//      below the cursor's byte); a request for piece k needs slot k mod 3 free, i.e. k + 3 > the cursor's piece.
// NS: ring slots of the cadenced form.  3 (above); or 2 with 64-byte pieces on the four-step cadence: a lane requests the piece below its
//     cursor's at the first group top after the cursor entered a piece (the piece above is dead from then on: the window only looks down),
//     at most 22 bytes in; it is parked one group later, at most 44 bytes in -- before the window (29 bytes of reach per group) or the
//     cursor gets there.  136 bytes of ring a lane instead of 200.
// UA: a stream's 64-byte bursts start where the stream's output starts, not at the next 64-byte line (small batches: the first burst is
//     then a full one like the others instead of a masked round, ~5 us of a lane's ~80; the lines that straddle cost nothing there)
template <bool SO, int WG = 1, int PLOG = 4, bool CAD = false, int NS = 3, bool UA = false, bool QW = false>
__device__ __forceinline__ void huf0_stream_body(const uint8_t* __restrict__ blocks, const uint64_t* __restrict__ boffs,
                                                 uint64_t nchunks, uint8_t* __restrict__ out,
                                                 const uint64_t* __restrict__ ooffs, int64_t* __restrict__ rets,
                                                 const uint8_t* __restrict__ desc, const uint8_t* __restrict__ share)
{
    static_assert(SO ? (WG == 1 || WG == 2 || WG == 4) : WG == 1, "a workgroup of the one-table kernel stays inside one 64-chunk segment");
    static_assert(!CAD || PLOG == 5 || PLOG == 6, "the cadenced refill: 32-byte pieces every four steps or 64-byte pieces every eight");
    static_assert(NS == 3 || (NS == 2 && CAD && PLOG == 6), "two ring slots: 64-byte pieces");
    static_assert(!QW || (CAD && NS == 3 && PLOG == 6 && !UA), "quad-wide requests and paired bursts: the cadenced three-slot form with 64-byte pieces and line-aligned bursts");
    constexpr int kThreads = 64 * WG, kChunks = kThreads / 4;
    if ((share[(uint64_t)blockIdx.x * kChunks >> 6] != 0) != SO) return;      // the other instantiation's
    auto sync = [] { if constexpr (SO) __syncthreads(); else wave_sync(); };
    __shared__ __attribute__((aligned(16))) uint8_t s_c[SO ? 336 + (2u << kSharedMaxLog) : 16 * kCStride];
    constexpr int kPLog = PLOG, kPB = 1 << kPLog, kPL = kPB / 16, kRingStride = CAD && NS == 3 ? 3 * (1 << PLOG) + 8 : ring_stride(PLOG);
#ifndef HUF0_CAD_PAD
#define HUF0_CAD_PAD 0                    // experiment: extra LDS bytes a workgroup of the cadenced kernel claims (fewer resident waves)
#endif
    __shared__ __attribute__((aligned(16))) uint8_t s_ring[kThreads * kRingStride + (CAD ? HUF0_CAD_PAD : 0)];
    const int t = threadIdx.x, q = t >> 2, j = t & 3;
    const uint64_t chunk0 = (uint64_t)blockIdx.x * kChunks;
    const uint64_t chunk = chunk0 + (uint64_t)q;
    const bool exists = chunk < nchunks;
    const uint64_t b0 = exists ? boffs[chunk] : 0, b1 = exists ? boffs[chunk + 1] : 0;
    const uint64_t o0 = exists ? ooffs[chunk] : 0, o1 = exists ? ooffs[chunk + 1] : 0;
    const uint8_t* const src = blocks + b0;
    uint8_t* const dst = out + o0;
    const uint64_t csize = b1 - b0, dsize = o1 - o0;
    uint8_t* const cbase = SO ? s_c : s_c + q * kCStride;       // (one tree for the wave: everybody reads the first chunk's copy)
    const uint8_t* const sorted = cbase;
    const uint32_t* const tab = (const uint32_t*)(cbase + 256);
    uint16_t* const table8 = (uint16_t*)(cbase + 320);

    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    typedef v4u __attribute__((aligned(1), may_alias)) v4u_a1;
    typedef v4u __attribute__((aligned(16), may_alias)) v4u_a16;
    if constexpr (SO) {                                           // the leader's descriptor stands for all 64
        if (t < 20) {
            const v4u v = *(const v4u_a16*)(desc + (chunk0 & ~(uint64_t)63) * kDescStride + 16u * (uint32_t)t);      // (the segment's leader: its followers hold no copies)
            uint32_t* const d = (uint32_t*)(s_c + 16u * (uint32_t)t);
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
    } else {
    // ---- descriptors of this wave's 16 chunks -> LDS: 320 pieces of 16 bytes, five per lane
#pragma unroll
    for (int r = 0; r < 5; r++) {
        const uint32_t piece = (uint32_t)r * 64 + (uint32_t)t, c = piece / 20u, part = piece % 20u;
        v4u v = {0, 0, 0, 0};
        if (chunk0 + c < nchunks) v = *(const v4u_a16*)(desc + (chunk0 + c) * kDescStride + 16u * part);
        uint32_t* const d = (uint32_t*)(s_c + c * kCStride + 16u * part);
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    }

    // ---- HUF_decompress's conventions (huf_decompress.c): stored, one repeated byte, or a coded block
    int mode = 0;                                                 // 0 nothing / damaged, 1 stored, 2 repeated byte, 3 coded
    int64_t ret = 0;
    if (exists) {
        if (dsize == 0) ret = csize == 0 ? 0 : kCorrupt;
        else if (b1 < b0 || o1 < o0 || csize == 0 || csize > dsize) ret = kCorrupt;
        else if (csize == dsize) mode = 1;
        else if (csize == 1) mode = 2;
        else mode = 3;
    }
    if (mode == 1) for (uint64_t k = (uint64_t)j; k < dsize; k += 4) dst[k] = src[k];
    if (mode == 2) { const uint8_t v = src[0]; for (uint64_t k = (uint64_t)j; k < dsize; k += 4) dst[k] = v; }
    sync();
    const uint32_t hl = mode == 3 ? (tab[0] & 0xffffu) : 0u, tl = mode == 3 ? (tab[0] >> 16) : 0u;
    if (mode == 3 && hl == 0) ret = kCorrupt;
    const bool coded = mode == 3 && hl != 0;

    // ---- table8: entry b = the code that starts with the 8 bits b, if it is at most 8 bits long.  Lane j of the
    // quad fills entries 64 j .. 64 j + 63; the weight only ever grows along them.
    // (SO = false: is this wave the other instantiation's?  Same test, on the LDS copies.)
    bool same_lane = !exists;
    if (exists && coded) {
        same_lane = true;
        const uint32_t* const mine = (const uint32_t*)cbase + 20 * j;
        const uint32_t* const first = (const uint32_t*)s_c + 20 * j;
#pragma unroll
        for (int k = 0; k < 20; k++) same_lane = same_lane && mine[k] == first[k];
    }
    const uint32_t tab00 = *(const uint32_t*)(s_c + 256);
    const bool shared = SO || (__ballot(!same_lane) == 0 && chunk0 < nchunks && (tab00 & 0xffffu) != 0u && __builtin_amdgcn_readfirstlane((int)coded) != 0);
    if (coded && !shared) {
        const uint32_t start_short = tl > 8u ? (tab[tl - 7u] & 0xffffu) : 0u;      // first index of the codes of <= 8 bits
        uint32_t w = 1, e = tab[1], nxt = tab[2] & 0xffffu;
        for (uint32_t k = 0; k < 64; k++) {
            const uint32_t b = 64u * (uint32_t)j + k;
            const uint32_t idx0 = tl >= 8u ? b << (tl - 8u) : b >> (8u - tl);
            while (w < 12u && idx0 >= nxt) { w++; e = tab[w]; nxt = w < 12u ? (tab[w + 1] & 0xffffu) : 0xffffffffu; }
            const uint32_t pos = (e >> 16) + ((idx0 - (e & 0xffffu)) >> (w - 1u));
            uint32_t entry;
            if (idx0 >= start_short) entry = sorted[pos & 0xffu] | ((tl + 1u - w) << 8);                    // the code is <= 8 bits: symbol | length << 8
            else if (idx0 + (1u << (tl - 8u)) <= nxt) entry = 0x8000u | ((w - 1u) << 8) | (pos & 0xffu);   // longer codes, all of weight w: where they start in `sorted`
            else entry = 0xffffu;                                                                         // codes of several weights share these 8 bits: the general search
            table8[b] = (uint16_t)entry;
        }
    }
    sync();

    // ---- One code for the whole wave: ONE full decode table (2^tableLog entries of symbol | length << 8) instead of sixteen
    // 8-bit prefix tables -- a symbol is then one LDS read and no weight logic at all.
    uint16_t* const full_table = (uint16_t*)(s_c + (SO ? 336 : kCStride + 12));      // 16-byte aligned; SO = false: over the other chunks' space
    sync();
    if (shared) {
        const uint32_t* const tab0 = (const uint32_t*)(s_c + 256);
        const uint32_t tl0 = tab0[0] >> 16, size = 1u << tl0, per = size >= (uint32_t)kThreads ? size / (uint32_t)kThreads : 1u;
        uint32_t w = 1, e = tab0[1], nxt = tab0[2] & 0xffffu;
        for (uint32_t k = 0; k < per; k++) {
            const uint32_t idx0 = (uint32_t)t * per + k;
            if (idx0 >= size) break;
            while (w < 12u && idx0 >= nxt) { w++; e = tab0[w]; nxt = w < 12u ? (tab0[w + 1] & 0xffffu) : 0xffffffffu; }
            const uint32_t pos = (e >> 16) + ((idx0 - (e & 0xffffu)) >> (w - 1u));
            full_table[idx0] = (uint16_t)((tl0 + 1u - w) | ((uint32_t)s_c[pos & 0xffu] << 8));      // length | symbol << 8: the length shifts the window as it is
        }
    }
    sync();

    // ---- lane j decodes stream j (HUF_decompress4X1_usingDTable_internal).  Set-up per lane, then ONE
    // wave-uniform loop: the quad exchanges of the output path need every lane, streamless ones included.
    bool bad = false;
    // long codes (> 8 bits) have weight <= tableLog - 8 <= 4: their weight is 1 + #{k in 2..4 : start[k] <= idx}
    const uint32_t E2 = coded ? tab[2] : 0xffffu, E3 = coded ? tab[3] : 0xffffu, E4 = coded ? tab[4] : 0xffffu;
    const uint32_t T2 = E2 & 0xffffu, T3 = E3 & 0xffffu, T4 = E4 & 0xffffu;
    const uint8_t* sp = blocks;                                   // this lane's stream
    int32_t P = 0;
    uint8_t* op = dst;
    uint64_t left = 0;
    if (coded) {
        const uint8_t* const ip = src + hl;
        const uint64_t n = csize - hl;
        if (n < 10) bad = true;
        uint64_t l[4] = {0, 0, 0, 0};
        if (!bad) {
            l[0] = (uint64_t)ip[0] | ((uint64_t)ip[1] << 8);
            l[1] = (uint64_t)ip[2] | ((uint64_t)ip[3] << 8);
            l[2] = (uint64_t)ip[4] | ((uint64_t)ip[5] << 8);
            if (6 + l[0] + l[1] + l[2] > n) bad = true;
            else l[3] = n - 6 - l[0] - l[1] - l[2];
            if (l[3] >= (1ull << 27)) bad = true;                 // the bit cursor is 32 bits: streams below 128 MiB (a Huff0 block is at most 128 KB)
        }
        if (!bad) {
            const uint64_t seg = (dsize + 3) / 4;
            uint64_t so = 6;
#pragma unroll
            for (int k = 0; k < 3; k++) so += k < j ? l[k] : 0;
            const uint64_t slen = j == 0 ? l[0] : j == 1 ? l[1] : j == 2 ? l[2] : l[3];
            uint64_t w0 = seg * (uint64_t)j;
            w0 = w0 < dsize ? w0 : dsize;
            const uint64_t w1 = j == 3 ? dsize : (w0 + seg < dsize ? w0 + seg : dsize);
            if (slen < 1 || ip[so + slen - 1] == 0) bad = true;
            if (!bad) {
                sp = ip + so;
                P = 8 * (int32_t)(slen - 1) + highbit(sp[slen - 1]);
                op = dst + w0;
                left = w1 - w0;
            }
        }
    }
    const bool streaming = left > 0 || (coded && !bad);           // has a stream whose end must be checked
    const uint32_t look_shift = 32u - (tl ? tl : 1u);
    const uint32_t bmask = tl > 8u ? (1u << (tl - 8u)) - 1u : 0u;     // the index bits below the 8-bit prefix
    const uint32_t t8 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)table8;
    const uint32_t so8 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)sorted;
    // The stream is read from its last byte down.  It reaches the lane as 16-byte aligned PIECES parked in a per-lane
    // LDS ring of two pieces (the 8-byte window never spans more); the piece below the ring is in flight in a register
    // and is only touched -- written into the ring slot the cursor just left -- when the cursor crosses into the next
    // piece, ~3 steps after it was requested.  (The first version rotated three pieces through registers: the
    // compiler's copies for the rotation made every step wait for the piece requested one step earlier.  A ring of
    // two 64-byte blocks hid the latency better but cost 8.7 KB of LDS a wave: 7 waves a CU instead of 10.)
    // Byte i of the stream sits at ring offset (s_al + i) & 31; piece k covers stream bytes [16 k - s_al, 16 k - s_al + 16).
    const uint32_t s_al = (uint32_t)((uintptr_t)sp & (uint32_t)(kPB - 1));
    // An aligned piece may start before `blocks` or end behind the last block by less than its size: it never leaves
    // the page of a byte that IS in the buffer, and those bytes are never looked at.
    const uint8_t* const sp_al = sp - s_al;
    const int32_t last_piece = streaming ? (int32_t)((((P > 0 ? (uint32_t)(P - 1) >> 3 : 0u)) + s_al) >> kPLog) : -1;
    struct Piece { v4u q[kPL]; };
    auto load_piece = [&](int32_t k) -> Piece {                   // pieces outside the stream read as zero, never touched
        Piece f;
#pragma unroll
        for (int i = 0; i < kPL; i++) f.q[i] = v4u{0, 0, 0, 0};
#ifdef ABL_NO_LOAD
        if (k < 0 || k > last_piece || k < last_piece - 2) return f;
#else
        if (k < 0 || k > last_piece) return f;
#endif
#pragma unroll
        for (int i = 0; i < kPL; i++) f.q[i] = *(const v4u_a16*)(sp_al + kPB * (int64_t)k + 16 * i);
        return f;
    };
    const uint32_t ring = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)s_ring + (uint32_t)t * kRingStride;
    auto park = [&](int32_t pk, const Piece& f) {                 // piece pk -> its ring slot
        const bool s0 = ((uint32_t)pk & 1u) == 0;
        const uint32_t at = ring + (s0 ? 0u : (uint32_t)kPB);
        typedef __attribute__((address_space(3))) uint64_t lds_u64;
#pragma unroll
        for (int i = 0; i < kPL; i++) {
            *(lds_u64*)(uintptr_t)(at + 16u * i) = (uint64_t)f.q[i].x | ((uint64_t)f.q[i].y << 32);
            *(lds_u64*)(uintptr_t)(at + 16u * i + 8u) = (uint64_t)f.q[i].z | ((uint64_t)f.q[i].w << 32);
        }
        // the first 8 bytes of slot 0 again behind slot 1: a window that wraps reads straight on (fast_step).  Slot 1
        // rewrites its own last 8 bytes instead -- no branch.
        *(lds_u64*)(uintptr_t)(ring + 2u * kPB - (s0 ? 0u : 8u)) =
            s0 ? ((uint64_t)f.q[0].x | ((uint64_t)f.q[0].y << 32)) : ((uint64_t)f.q[kPL - 1].z | ((uint64_t)f.q[kPL - 1].w << 32));
    };
    int32_t cur_b = last_piece;                                   // piece of the cursor's byte
    Piece fl;
    // ---- CAD state: the ring holds pieces cur_b .. low_k in slots (piece mod 3); `pend` is the piece requested at the last group top
    // outside every descriptor (they end below 0xfffffe00): a load answers zeros, a store is dropped -- and so does kDrop + 16 j, j < 4: the quad's
    // lanes add their part's offset to a broadcast one without asking whether it is this one (a compare and a select less per request)
    constexpr uint32_t kDrop = 0xffffff00u;
    auto uniform64 = [](uint64_t v) {
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    };
    const uint64_t wave_first = chunk0 + (uint64_t)(((uint32_t)t >> 6) << 4);
    const uint64_t wf = wave_first < nchunks ? wave_first : nchunks;
    const uint64_t in_base = CAD ? uniform64(((uint64_t)(uintptr_t)blocks + boffs[wf]) & ~(uint64_t)31) : 0;
    const uint64_t in_span = CAD ? uniform64((((uint64_t)(uintptr_t)blocks + boffs[nchunks]) - in_base + 15) & ~(uint64_t)15) : 0;
    const uint64_t out_base = CAD ? uniform64(((uint64_t)(uintptr_t)out + ooffs[wf]) & ~(uint64_t)15) : 0;
    const uint64_t out_span = CAD ? uniform64(((uint64_t)(uintptr_t)out + ooffs[nchunks]) - out_base) : 0;
    const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)in_base, 0, (uint32_t)(in_span < 0xfffffe00ull ? in_span : 0xfffffe00ull), 0x00020000);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)out_base, 0, (uint32_t)(out_span < 0xfffffe00ull ? out_span : 0xfffffe00ull), 0x00020000);
    const uint32_t sp_off = CAD && streaming ? (uint32_t)((uint64_t)(uintptr_t)sp_al - in_base) : 0u;
    int32_t low_k = 0;                                            // lowest piece in the ring or in flight
    uint32_t cur_s32 = 0, m1_s32 = 0, low_s32 = 0, pend_s32 = 0;  // byte offset of the ring slot of cur_b / of cur_b - 1 / of low_k / of the piece in flight
    bool pend_on = false;
    Piece pend;
    auto slot_below = [](uint32_t s32) { return s32 == 0u ? 2u * (uint32_t)kPB : s32 - (uint32_t)kPB; };
    auto park3 = [&](uint32_t s32, const Piece& f) {               // a 32-byte piece -> ring slot s32 / 32 (+ the first 8 bytes of slot 0 again behind slot 2)
        typedef __attribute__((address_space(3))) uint64_t lds_u64;
        const uint32_t at = ring + s32;
#pragma unroll
        for (int i = 0; i < kPL; i++) {
            *(lds_u64*)(uintptr_t)(at + 16u * i) = (uint64_t)f.q[i].x | ((uint64_t)f.q[i].y << 32);
            *(lds_u64*)(uintptr_t)(at + 16u * i + 8u) = (uint64_t)f.q[i].z | ((uint64_t)f.q[i].w << 32);
        }
        if (s32 == 0u) *(lds_u64*)(uintptr_t)(ring + 3u * kPB) = (uint64_t)f.q[0].x | ((uint64_t)f.q[0].y << 32);
    };
    // QW (round 5): the QUAD fetches and parks.  A lane asking for its own 64-byte piece with four 16-byte requests makes a wave's load 64 requests
    // to 64 different lines, four times over; the memory system takes this kernel's lane-wise requests + 64-byte bursts at 2.2 TB/s -- the probe
    // tools/probes/huf0_pattern.hip replays the pattern WITHOUT any arithmetic in 2.6 ms, the stage's own time -- and quad-wide requests with
    // 128-byte bursts at 4.8 TB/s (1.19 ms).  So the four lanes of a quad (= the four streams of a chunk) fetch each of their streams' pieces
    // together: lane j asks for bytes 16 j .. 16 j + 15 of stream s's piece, s = 0 .. 3 (the offset comes from lane s over the quad's DPP
    // network) -- one 64-byte request a quad, the same four load instructions -- and writes its 16 bytes into stream s's ring slot itself (LDS is
    // everybody's: no transpose).  A lane that asks for nothing passes kDrop (the load answers zeros, no traffic) / ~0 (nothing parked).
    auto quad_bcast = [](uint32_t v, int s) -> uint32_t {
        return (uint32_t)(s == 0 ? __builtin_amdgcn_mov_dpp((int)v, 0x00, 0xf, 0xf, true) : s == 1 ? __builtin_amdgcn_mov_dpp((int)v, 0x55, 0xf, 0xf, true)
                          : s == 2 ? __builtin_amdgcn_mov_dpp((int)v, 0xAA, 0xf, 0xf, true) : __builtin_amdgcn_mov_dpp((int)v, 0xFF, 0xf, 0xf, true));
    };
    auto quad_fetch = [&](uint32_t vo, Piece& pc) {               // vo: byte offset of this lane's piece in brsrc, or kDrop; pc.q[s] <- bytes 16 j .. of stream s's piece
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const uint32_t v = quad_bcast(vo, s);
            const auto t0 = __builtin_amdgcn_raw_buffer_load_b128(brsrc, v + 16u * (uint32_t)j, 0, 0);
            pc.q[s] = v4u{t0[0], t0[1], t0[2], t0[3]};
        }
    };
    auto quad_park = [&](uint32_t slot32, const Piece& pc) {      // slot32: byte offset of the ring slot this lane's piece goes to, or ~0
        typedef __attribute__((address_space(3))) uint64_t lds_u64;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const uint32_t d = quad_bcast(slot32, s);
            if (d != ~0u) {
                const uint32_t rs = ring + (uint32_t)((s - j) * kRingStride);      // stream s's ring
                const uint32_t at = rs + d + 16u * (uint32_t)j;
                *(lds_u64*)(uintptr_t)at = (uint64_t)pc.q[s].x | ((uint64_t)pc.q[s].y << 32);
                *(lds_u64*)(uintptr_t)(at + 8u) = (uint64_t)pc.q[s].z | ((uint64_t)pc.q[s].w << 32);
                if (d == 0u && j == 0) *(lds_u64*)(uintptr_t)(rs + 3u * kPB) = (uint64_t)pc.q[s].x | ((uint64_t)pc.q[s].y << 32);   // slot 0's head again behind slot 2
            }
        }
    };
    // the top of every group of four steps.  Preal: the lane's true cursor (a lane that rides along carries a parked one in P)
    // (64-byte pieces: every second group -- eight steps take at most 44 bytes)
    int32_t pend_k = 0;
    auto group_top = [&](int g, int32_t Preal) {
        if constexpr (CAD && NS == 2) {
            if (pend_on) { park(pend_k, pend); low_k = pend_k; }
            const uint32_t xr = ((uint32_t)(Preal > 0 ? Preal - 1 : 0) >> 3) + s_al;
            if (streaming && (int32_t)(xr >> kPLog) < cur_b) cur_b--;
            const int32_t k = cur_b - 1;
            const bool want = streaming && low_k == cur_b && k >= 0;
            const uint32_t vo = want ? sp_off + (uint32_t)kPB * (uint32_t)k : kDrop;
#pragma unroll
            for (int i = 0; i < kPL; i++) {
                const auto t0 = __builtin_amdgcn_raw_buffer_load_b128(brsrc, want ? vo + 16u * i : kDrop, 0, 0);
                pend.q[i] = v4u{t0[0], t0[1], t0[2], t0[3]};
            }
            pend_on = want;
            pend_k = k;
        } else if constexpr (CAD) {
            if (g % (kPB / 32) != 0) return;
            if constexpr (QW) quad_park(pend_on ? pend_s32 : ~0u, pend); else
            if (pend_on) park3(pend_s32, pend);
            const uint32_t xr = ((uint32_t)(Preal > 0 ? Preal - 1 : 0) >> 3) + s_al;
            if (streaming && (int32_t)(xr >> kPLog) < cur_b) {   // at most one boundary per cadence
                cur_b--;
                cur_s32 = slot_below(cur_s32);
            }
            m1_s32 = slot_below(cur_s32);
            const int32_t k = low_k - 1;
#ifdef ABL_CAD_NO_LOAD
            const bool want = streaming && k >= 0 && k + 3 > cur_b && k > last_piece;      // ablation: never
#else
            const bool want = streaming && k >= 0 && k + 3 > cur_b;
#endif
            const uint32_t vo = want ? sp_off + (uint32_t)kPB * (uint32_t)k : kDrop;
            if constexpr (QW) quad_fetch(vo, pend); else
#pragma unroll
            for (int i = 0; i < kPL; i++) {
                const auto t0 = __builtin_amdgcn_raw_buffer_load_b128(brsrc, want ? vo + 16u * i : kDrop, 0, 0);
                pend.q[i] = v4u{t0[0], t0[1], t0[2], t0[3]};
            }
            pend_on = want;
            pend_s32 = slot_below(low_s32);
            if (want) { low_k = k; low_s32 = pend_s32; }
        }
    };
    if constexpr (CAD && NS == 2) {
        const Piece f0 = load_piece(cur_b), f1 = load_piece(cur_b - 1);
        park(cur_b, f0);
        park(cur_b - 1, f1);
        low_k = cur_b - 1;
        fl = f0;                                                  // (unused in this form)
    } else if constexpr (CAD && QW) {
        cur_s32 = (uint32_t)kPB * ((uint32_t)(cur_b < 0 ? 0 : cur_b) % 3u);
        m1_s32 = slot_below(cur_s32);
        Piece f0, f1;                                             // (pieces outside the stream are not fetched: what the ring holds there is never consumed by a valid stream)
        const bool w0 = streaming && cur_b >= 0, w1 = streaming && cur_b >= 1;
        quad_fetch(w0 ? sp_off + (uint32_t)kPB * (uint32_t)cur_b : kDrop, f0);
        quad_fetch(w1 ? sp_off + (uint32_t)kPB * (uint32_t)(cur_b - 1) : kDrop, f1);
        quad_park(w0 ? cur_s32 : ~0u, f0);
        quad_park(w1 ? m1_s32 : ~0u, f1);
        low_k = cur_b - 1;
        low_s32 = m1_s32;
        fl = f0;                                                  // (unused in this form)
    } else if constexpr (CAD) {
        const Piece f0 = load_piece(cur_b), f1 = load_piece(cur_b - 1);
        cur_s32 = (uint32_t)kPB * ((uint32_t)(cur_b < 0 ? 0 : cur_b) % 3u);
        m1_s32 = slot_below(cur_s32);
        park3(cur_s32, f0);
        park3(m1_s32, f1);
        low_k = cur_b - 1;
        low_s32 = m1_s32;
        fl = f0;                                                  // (unused in this form)
    } else {
        const Piece f0 = load_piece(cur_b), f1 = load_piece(cur_b - 1);
        fl = load_piece(cur_b - 2);
        park(cur_b, f0);
        park(cur_b - 1, f1);
    }
    sync();
    typedef __attribute__((address_space(3))) const uint16_t lds_u16;
    typedef __attribute__((address_space(3))) const uint8_t lds_u8c;
    typedef __attribute__((address_space(3))) const uint32_t lds_u32c;
    // one step = up to 4 symbols = one dword of output; branch-free except for the block crossing
    const uint32_t ft = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)full_table;
    auto step = [&](uint32_t m, auto SH) -> uint32_t {
        constexpr bool kShared = decltype(SH)::value;
        const int32_t Pc = P;
        const uint32_t x = ((uint32_t)(Pc > 0 ? Pc - 1 : 0) >> 3) + s_al;      // the cursor's byte, counted from piece 0
        uint32_t o, d0, d1, d2;
        if constexpr (CAD && NS == 3) {                           // the cursor's piece is the group top's or the one below: its slot is known
            const uint32_t sb = (int32_t)(x >> kPLog) == cur_b ? cur_s32 : m1_s32;
            int32_t oo = (int32_t)(sb + (x & (uint32_t)(kPB - 1))) - 7;
            oo += oo < 0 ? 3 * kPB : 0;                           // below slot 0 is the end of slot 2; past slot 2 is the copy of slot 0's head
            o = (uint32_t)oo;
            const uint32_t a = ring + (o & ~3u);
            d0 = *(lds_u32c*)(uintptr_t)a; d1 = *(lds_u32c*)(uintptr_t)(a + 4u); d2 = *(lds_u32c*)(uintptr_t)(a + 8u);
        } else {
        if constexpr (!CAD) {
        if ((int32_t)(x >> kPLog) < cur_b) {                      // crossed into the piece below: the one in flight takes the freed slot
            park(cur_b - 2, fl);
            cur_b--;
            fl = load_piece(cur_b - 2);
        }
        }
        // the 8 bytes ending at byte x: three aligned dwords of the ring, two v_alignbyte
        o = x - 7u;                                               // may be "negative": bytes before the stream read as what the ring holds, masked below
        constexpr uint32_t kRM = 2u * kPB - 4u;
        d0 = *(lds_u32c*)(uintptr_t)(ring + (o & kRM));
        d1 = *(lds_u32c*)(uintptr_t)(ring + ((o + 4u) & kRM));
        d2 = *(lds_u32c*)(uintptr_t)(ring + ((o + 8u) & kRM));
        }
        const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, o & 3u), hi0 = __builtin_amdgcn_alignbyte(d2, d1, o & 3u);
        uint64_t win = (((uint64_t)hi0 << 32) | lo) << (7 - (int)((uint32_t)(Pc - 1) & 7u));
        if (__ballot(Pc < 64) != 0) {                             // only the last steps of a stream: nothing before its first bit
            const uint64_t keep = Pc >= 64 ? ~0ull : (Pc > 0 ? ~0ull << (64 - Pc) : 0ull);
            win &= keep;
        }
        uint32_t word = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t hi = (uint32_t)(win >> 32);
            if constexpr (kShared) {
                const uint32_t e = *(lds_u16*)(uintptr_t)(ft + ((hi >> look_shift) << 1));
                const bool on = (uint32_t)k < m;
                const uint32_t nb = on ? (e & 0xffu) : 0u;
                word |= (on ? (e >> 8) : 0u) << (8 * k);
                win <<= nb;
                P -= (int32_t)nb;
                continue;
            }
            const uint32_t e8 = *(lds_u16*)(uintptr_t)(t8 + ((hi >> 24) << 1));
            const uint32_t idx = hi >> look_shift;
            // a longer code whose 8-bit prefix holds one weight only: its symbol sits at base + (low index bits >> (w - 1))
            uint32_t wm1 = (e8 >> 8) & 3u;
            uint32_t pos = (e8 & 0xffu) + ((idx & bmask) >> wm1);
            if (__ballot(e8 == 0xffffu) != 0) {                   // rare: prefixes shared by several weights -- weight = 1 + #{k in 2..4 : start[k] <= idx}
                const bool g2 = idx >= T2, g3 = idx >= T3, g4 = idx >= T4;
                const uint32_t el = g4 ? E4 : (g3 ? E3 : (g2 ? E2 : 0u));
                const uint32_t wg = (uint32_t)g2 + (uint32_t)g3 + (uint32_t)g4;
                const bool mixed = e8 == 0xffffu;
                pos = mixed ? (el >> 16) + ((idx - (el & 0xffffu)) >> wg) : pos;
                wm1 = mixed ? wg : wm1;
            }
            const uint32_t syl = *(lds_u8c*)(uintptr_t)(so8 + (pos & 0xffu));
            const bool is_short = (e8 & 0x8000u) == 0;
            const bool on = (uint32_t)k < m;
            const uint32_t sym = is_short ? (e8 & 0xffu) : syl;
            const uint32_t nb = on ? (is_short ? (e8 >> 8) : (tl - wm1)) : 0u;
            word |= (on ? sym : 0u) << (8 * k);
            win <<= nb;
            P -= (int32_t)nb;
        }
        return word;
    };
    // The bulk of a stream: every lane of the wave has >= 64 symbols to go and its cursor far enough from the stream's
    // first bit that 16 steps (<= 16 * 4 * 12 bits) need no masking, no symbol count and no bounds on the piece index.
    // 4 symbols are then ~45 instructions instead of ~130 (the per-symbol `on` masks alone took 40 SGPRs a step).
    constexpr int32_t kFastBits = 64 + 3 * 48;                    // four steps at a time
    // Round 5: the CARRIED window (one-table path, 64-byte pieces on the cadence).  A step used to start by fetching its 64-bit window
    // from the ring at the cursor's byte -- address arithmetic from P, an LDS round trip, two v_alignbyte and a 64-bit shift, all of it on the
    // lane's serial chain in front of the first look-up: five dependent LDS round trips per four symbols.  But a step leaves >= 64 - 4 * 12
    // = 16 valid bits in the window it had, which is all its successor's FIRST look-up needs.  So the window is carried from step to step
    // (valid while carry_P == P), and what a step fetches from the ring is the 64 bits BELOW its window (from bit P - 65 down), needed only
    // when the step ends: win' = win << c | below >> (64 - c).  The fetch and its arithmetic leave the chain (four round trips per four
    // symbols, ~7 dependent VALU instructions a step less); the LDS reads are the same in number.  Reach: the fetch looks 8 bytes further
    // down than the window did: eight steps take <= 48 bytes, + 7 (window) + 8 = 63 <= the 64 bytes of piece cur_b - 1 that the cadence
    // keeps resident below the group top's cursor (two slots, refilled every four steps: a piece is requested <= 22 bytes after the cursor entered
    // the one above it and parked <= 44 bytes in; until then the steps read <= 38.5 + 15 = 54 bytes into that piece).  Lanes that ride along (parked cursor), bits below a stream's first one: garbage that is
    // never consumed, every address inside the lane's ring as before.
    constexpr bool kCarry = CAD && PLOG == 6 && HUF0_CARRY_WINDOW;
    uint64_t carry_pre = 0, carry_fill = 0;                       // the window = carry_pre | carry_fill (the first look-up reads carry_pre alone)
    int32_t carry_P = -1;                                         // the cursor the carried window belongs to
    auto ring_bytes = [&](uint32_t top) -> uint64_t {             // CAD, NS == 3: the 8 stream bytes that end with the byte of bit `top` (that byte on top)
        const uint32_t x = (top >> 3) + s_al;
        uint32_t o, a;
        if constexpr (NS == 3) {
            const uint32_t sb = (int32_t)(x >> kPLog) == cur_b ? cur_s32 : m1_s32;
            int32_t oo = (int32_t)(sb + (x & (uint32_t)(kPB - 1))) - 7;
            oo += oo < 0 ? 3 * kPB : 0;
            o = (uint32_t)oo;
            a = ring + (o & ~3u);
        } else {                                                  // two slots: piece k sits in slot k & 1, byte i of the stream at (s_al + i) & 127
            o = x - 7u;
            a = ring + (o & (2u * kPB - 4u));
        }
        const uint32_t d0 = *(lds_u32c*)(uintptr_t)a, d1 = *(lds_u32c*)(uintptr_t)(a + 4u), d2 = *(lds_u32c*)(uintptr_t)(a + 8u);
        const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, o & 3u), hi0 = __builtin_amdgcn_alignbyte(d2, d1, o & 3u);
        return ((uint64_t)hi0 << 32) | lo;
    };
    uint32_t carry_rb7 = 0;                                       // ring bit position of the bits below the carried window (fast_step)
    auto ring_at = [&](uint32_t y) -> uint64_t {                  // the 8 bytes at ring offsets y .. y + 7 (y < 192: past slot 2 sits the copy of slot 0's head)
        const uint32_t a = ring + (y & ~3u);
        const uint32_t d0 = *(lds_u32c*)(uintptr_t)a, d1 = *(lds_u32c*)(uintptr_t)(a + 4u), d2 = *(lds_u32c*)(uintptr_t)(a + 8u);
        const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, y & 3u), hi0 = __builtin_amdgcn_alignbyte(d2, d1, y & 3u);
        return ((uint64_t)hi0 << 32) | lo;
    };
    // keep = false (a wave's first and last rounds): the lane goes through the motions and leaves its cursor and its window where they were
    auto fast_step = [&](auto SH, bool keep = true) -> uint32_t {
        constexpr bool kShared = decltype(SH)::value;
        if constexpr (kCarry && kShared) {
            uint32_t sh;                                          // bits of the window's top byte that are above the cursor
            uint64_t below_bytes;
            if constexpr (NS == 3 && HUF0_CARRY_RINGPOS) {
                // Where the bits below sit in the ring is carried along as well: a piece k lives in slot k mod 3, so byte x of the stream (counted
                // from piece 0) is at ring offset x mod 192 -- no slot to look up, and a step that takes c bits moves the position by exactly c
                // bits modulo 8 * 192.  carry_rb7 = 8 * ((x - 7) mod 192) + bit, x the byte of bit P - 65: 8 instructions a step for address,
                // alignment and shift count instead of 17 (piece of x, compare with the cursor's, slot select, - 7, wrap, shift count from P).
                if (__ballot(carry_P != P) != 0) {                // after a masked step, a parked cursor, a round's first step: fetch it (wave-uniform, rare)
                    const uint32_t top = (uint32_t)P - 65u, x = (top >> 3) + s_al;
                    const uint32_t sb = (int32_t)(x >> kPLog) == cur_b ? cur_s32 : m1_s32;
                    int32_t oo = (int32_t)(sb + (x & (uint32_t)(kPB - 1))) - 7;
                    oo += oo < 0 ? 3 * kPB : 0;
                    carry_rb7 = 8u * (uint32_t)oo + (top & 7u);
                    const uint32_t sh0 = 7u - (top & 7u);
                    carry_pre = ring_bytes((uint32_t)P - 1u) << sh0;
                    carry_fill = (ring_at((uint32_t)oo) >> 1) >> (63u - sh0);
                }
                sh = 7u - (carry_rb7 & 7u);
                below_bytes = ring_at(carry_rb7 >> 3);
            } else {
                sh = 7u - (((uint32_t)P - 1u) & 7u);
                below_bytes = ring_bytes((uint32_t)P - 65u);
                if (__ballot(carry_P != P) != 0) {
                    carry_pre = ring_bytes((uint32_t)P - 1u) << sh;   // (its low `sh` bits are the top bits of the bytes below)
                    carry_fill = (below_bytes >> 1) >> (63u - sh);
                }
            }
            const uint64_t below = below_bytes << sh;             // the stream from bit P - 65 down
            const uint32_t e0 = *(lds_u16*)(uintptr_t)(ft + (((uint32_t)(carry_pre >> 32) >> look_shift) << 1));
            uint64_t win = (carry_pre | carry_fill) << (e0 & 63u);
            const uint32_t e1 = *(lds_u16*)(uintptr_t)(ft + (((uint32_t)(win >> 32) >> look_shift) << 1));
            win <<= e1 & 63u;
            const uint32_t e2 = *(lds_u16*)(uintptr_t)(ft + (((uint32_t)(win >> 32) >> look_shift) << 1));
            win <<= e2 & 63u;
            const uint32_t e3 = *(lds_u16*)(uintptr_t)(ft + (((uint32_t)(win >> 32) >> look_shift) << 1));
            const uint32_t c = (e0 + e1 + e2 + e3) & 0xffu;       // 4 .. 48 bits
            carry_pre = keep ? win << (e3 & 63u) : carry_pre;
            carry_fill = keep ? below >> ((64u - c) & 63u) : carry_fill;
            P -= keep ? (int32_t)c : 0;
            carry_P = P;                                          // (valid either way: the window at the top of the step was this lane's, fetched or carried)
            if constexpr (NS == 3 && HUF0_CARRY_RINGPOS) {
                const uint32_t r1 = carry_rb7 - c, r2 = r1 + 8u * 3u * (uint32_t)kPB;      // c bits down, modulo the ring (r1 wraps to a huge number below 0)
                carry_rb7 = keep ? (r1 < r2 ? r1 : r2) : carry_rb7;
            }
            const uint32_t w01 = __builtin_amdgcn_perm(e1, e0, 0x0c0c0501u), w23 = __builtin_amdgcn_perm(e3, e2, 0x0c0c0501u);
            return __builtin_amdgcn_perm(w23, w01, 0x05040100u);
        }
        const uint32_t pm1 = (uint32_t)P - 1u;
        const uint32_t x = (pm1 >> 3) + s_al;
        uint32_t o, a;
        if constexpr (CAD && NS == 3) {
            const uint32_t sb = (int32_t)(x >> kPLog) == cur_b ? cur_s32 : m1_s32;
            int32_t oo = (int32_t)(sb + (x & (uint32_t)(kPB - 1))) - 7;
            oo += oo < 0 ? 3 * kPB : 0;
            o = (uint32_t)oo;
            a = ring + (o & ~3u);
        } else {
        if constexpr (!CAD) {
        if ((int32_t)(x >> kPLog) < cur_b) {
            park(cur_b - 2, fl);
            cur_b--;
            fl = load_piece(cur_b - 2);
        }
        }
        o = x - 7u;
        a = ring + (o & (2u * kPB - 4u));
        }
        const uint32_t d0 = *(lds_u32c*)(uintptr_t)a, d1 = *(lds_u32c*)(uintptr_t)(a + 4u), d2 = *(lds_u32c*)(uintptr_t)(a + 8u);
        const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, o & 3u), hi0 = __builtin_amdgcn_alignbyte(d2, d1, o & 3u);
        uint64_t win = (((uint64_t)hi0 << 32) | lo) << (7u - (pm1 & 7u));
        if constexpr (!kShared) {                                 // the chunk's own 8-bit prefix table, then `sorted` for the longer codes
            uint32_t sy[4], nbs = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t hi = (uint32_t)(win >> 32);
                const uint32_t e8 = *(lds_u16*)(uintptr_t)(t8 + ((hi >> 24) << 1));
                const uint32_t idx = hi >> look_shift;
                uint32_t wm1 = (e8 >> 8) & 3u;
                uint32_t pos = (e8 & 0xffu) + ((idx & bmask) >> wm1);
                if (__ballot(e8 == 0xffffu) != 0) {
                    const bool g2 = idx >= T2, g3 = idx >= T3, g4 = idx >= T4;
                    const uint32_t el = g4 ? E4 : (g3 ? E3 : (g2 ? E2 : 0u));
                    const uint32_t wg = (uint32_t)g2 + (uint32_t)g3 + (uint32_t)g4;
                    const bool mixed = e8 == 0xffffu;
                    pos = mixed ? (el >> 16) + ((idx - (el & 0xffffu)) >> wg) : pos;
                    wm1 = mixed ? wg : wm1;
                }
                const uint32_t syl = *(lds_u8c*)(uintptr_t)(so8 + (pos & 0xffu));
                const bool is_short = (e8 & 0x8000u) == 0;
                sy[k] = is_short ? e8 : syl;
                const uint32_t nb = is_short ? (e8 >> 8) : (tl - wm1);
                win <<= nb;
                nbs += nb;
            }
            P -= (int32_t)nbs;
            const uint32_t w01 = __builtin_amdgcn_perm(sy[1], sy[0], 0x0c0c0400u), w23 = __builtin_amdgcn_perm(sy[3], sy[2], 0x0c0c0400u);
            return __builtin_amdgcn_perm(w23, w01, 0x05040100u);
        }
        const uint32_t e0 = *(lds_u16*)(uintptr_t)(ft + (((uint32_t)(win >> 32) >> look_shift) << 1));
        // (an entry is length | symbol << 8: the 64-bit shift takes its count from the low six bits of the entry as it is -- one instruction
        //  less on the chain of every symbol; the four lengths are the low byte of the entries' sum, the symbols their second bytes)
        win <<= e0 & 63u;
        const uint32_t e1 = *(lds_u16*)(uintptr_t)(ft + (((uint32_t)(win >> 32) >> look_shift) << 1));
        win <<= e1 & 63u;
        const uint32_t e2 = *(lds_u16*)(uintptr_t)(ft + (((uint32_t)(win >> 32) >> look_shift) << 1));
        win <<= e2 & 63u;
        const uint32_t e3 = *(lds_u16*)(uintptr_t)(ft + (((uint32_t)(win >> 32) >> look_shift) << 1));
        P -= (int32_t)((e0 + e1 + e2 + e3) & 0xffu);
        const uint32_t w01 = __builtin_amdgcn_perm(e1, e0, 0x0c0c0501u), w23 = __builtin_amdgcn_perm(e3, e2, 0x0c0c0501u);
        return __builtin_amdgcn_perm(w23, w01, 0x05040100u);
    };
    // The output leaves 64 bytes at a time: a lane collects 16 steps in registers, the quad transposes
    // its 16-byte pieces (two DPP butterfly stages) and every store writes ONE stream's 64 contiguous
    // bytes (4-byte stores per lane per step -- 524 288 open lines at the headline shape -- made this
    // phase 3.2 ms of a 3.7 ms launch).  A stream's final partial burst goes out narrow.
    const bool odd1 = (t & 1) != 0, odd2 = (t & 2) != 0;
    const uint32_t part = (uint32_t)t & 3u;
    uint32_t head = streaming && !UA ? (uint32_t)(0u - (uint32_t)(uintptr_t)op) & 63u : 0u;
    uint32_t hold[4][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};      // QW: stream qq's waiting lower half, this lane's 16 bytes of it
    bool held = false;                                            // QW: this lane's stream has a lower half waiting, at hold_off
    uint32_t hold_off = 0;
    auto run = [&](auto SH) {
    for (;;) {
        if (__ballot(left > 0) == 0) break;
        // a stream's first burst only goes up to the next 64-byte line of the output: every later one stores whole lines
        const bool head_lim = head != 0 && head < left;
        const uint64_t lim = head_lim ? (uint64_t)head : left;
        const bool full = lim >= 64;
        uint32_t wb[16];
#if HUF0_SPECULATIVE_TAIL
        // Round 3.  A round is 64 symbols a lane.  Masked steps (~130 instructions instead of ~50) are only needed where a lane must
        // stop after fewer than 64 symbols: a stream's FIRST burst (cut at the next 64-byte line of the output: the same round for
        // all 64 lanes) and its LAST.  The last bursts are made to fall into ONE round too: a lane that is down to its partial last
        // burst -- or done -- rides along through the others' full rounds from a parked cursor, its own put back after each, and when no lane has a full burst left one masked round
        // finishes them all.  (Before: one lane within 64 symbols or 208 bits of its end held all 64 in the masked path -- the ~17 %
        // of a wave's symbols between the end of its shortest and of its longest stream.)
        const bool live = left > 0;
        const bool burst = live && full && !head_lim;             // this lane decodes a whole 64-symbol burst this round
        const bool fast_round = __ballot(streaming && live && head_lim) == 0 && __ballot(burst) != 0;
        // (a lane that rides along decodes from cursor 0: `x` below is then far above any piece index, so its ring stays as it is)
        const int32_t P0 = P;
        if (fast_round && !burst) { P = 0; carry_P = 0; carry_rb7 = 0; }         // (its carried window is garbage like everything it decodes: no refetch for the riders' sake)
#else
        constexpr bool fast_round = false;
        const bool burst = full;
#endif
        bool last_round_done = false;
        if constexpr ((UA || (HUF0_FAST_TAILS && NS == 3)) && HUF0_SPECULATIVE_TAIL) {
            // The last round of a wave (no lane has 64 symbols left), small batches: a lane's whole steps of four symbols run as fast
            // steps too -- a lane past its last whole step takes its cursor back after each -- and ONE masked step behind them decodes
            // every lane's last 0 .. 3 symbols: 16 fast + 1 masked instead of 16 masked steps (~9 us -> ~5 of a lane's ~75).
            if (!fast_round) {
                const uint32_t lim64 = lim < 64 ? (uint32_t)lim : 64u;           // (a round that is not a fast one: the wave's last -- and, where the bursts are cut
                const uint32_t nfull = streaming ? lim64 >> 2 : 0u;               //  at the output's 64-byte lines, its first, in which a lane takes 0 .. 64 symbols)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    group_top(g, P);
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int32_t Pk = P;
                        wb[4 * g + i] = fast_step(SH, (uint32_t)(4 * g + i) < nfull);
                        P = (uint32_t)(4 * g + i) < nfull ? P : Pk;
                    }
                }
                group_top(0, P);
                const uint32_t m = (streaming && P >= -64) ? lim64 & 3u : 0u;
                const uint32_t wl = step(m, SH);
#pragma unroll
                for (int k = 0; k < 16; k++) wb[k] = (uint32_t)k == nfull ? wl : wb[k];
                last_round_done = true;
            }
        }
        if (!last_round_done) {
            const bool all_full = __ballot(streaming && lim < 64) == 0;      // (lanes without a stream just go through the motions)
#pragma unroll
            for (int g = 0; g < 4; g++) {
#if HUF0_SPECULATIVE_TAIL
                group_top(g, fast_round && !burst ? P0 : P);
#else
                group_top(g, P);
#endif
                if (fast_round || (all_full && __ballot(streaming && P < kFastBits) == 0)) {
#pragma unroll
                    for (int i = 0; i < 4; i++) wb[4 * g + i] = fast_step(SH);
                } else {                                          // the ends of the streams: rolled, so that the bulk's registers set the occupancy
                    uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
#pragma unroll 1
                    for (int i = 0; i < 4; i++) {
                        const uint32_t done = 16u * g + 4u * i;
                        const uint32_t m = (lim > done && P >= -64) ? (lim - done < 4 ? (uint32_t)(lim - done) : 4u) : 0u;
                        const uint32_t w = step(m, SH);
                        w0 = i == 0 ? w : w0; w1 = i == 1 ? w : w1; w2 = i == 2 ? w : w2; w3 = i == 3 ? w : w3;
                    }
                    wb[4 * g] = w0; wb[4 * g + 1] = w1; wb[4 * g + 2] = w2; wb[4 * g + 3] = w3;
                }
            }
        }
#if HUF0_SPECULATIVE_TAIL
        if (fast_round && !burst) P = P0;                         // rode along: the cursor goes back, nothing is kept
#endif
        const uint32_t cnt = fast_round ? (burst ? 64u : 0u) : (lim < 64 ? (uint32_t)lim : 64u);
        const bool full_out = fast_round ? burst : full;          // stores whole 64-byte lines this round
        if (!fast_round) head = 0;
        uint32_t v[4][4];
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int d = 0; d < 4; d++) v[k][d] = wb[4 * k + d];
#pragma unroll
        for (int k = 0; k < 4; k += 2)
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const uint32_t send = odd1 ? v[k][d] : v[k + 1][d];
                const uint32_t recv = (uint32_t)__builtin_amdgcn_mov_dpp((int)send, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
                if (odd1) v[k][d] = recv; else v[k + 1][d] = recv;
            }
#pragma unroll
        for (int k = 0; k < 2; k++)
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const uint32_t send = odd2 ? v[k][d] : v[k + 2][d];
                const uint32_t recv = (uint32_t)__builtin_amdgcn_mov_dpp((int)send, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
                if (odd2) v[k][d] = recv; else v[k + 2][d] = recv;
            }
        if constexpr (QW && HUF0_QW_PAIR) {
            // whole 128-byte lines: a burst that is a line's LOWER half waits in the quad's registers (16 bytes a lane, transposed already) for the
            // upper half one round later and the two leave back to back; a burst that is an upper half without a lower one waiting (a stream's
            // first) goes alone, and so does a lower half whose stream has no whole burst left (flushed in the first round its lane rides along,
            // or behind the loop).  Eight unconditional buffer stores a round, about half of them to a dropped offset.
            const uint32_t O = (uint32_t)((uint64_t)(uintptr_t)op - out_base);
            const bool upper = ((uint32_t)(uintptr_t)op & 64u) != 0;
            const uint32_t offA = held && (!full_out || upper) ? hold_off : kDrop;
            const uint32_t offB = full_out && upper ? O : kDrop;
            const bool take = full_out && !upper;
            typedef __attribute__((__vector_size__(4 * sizeof(uint32_t)))) uint32_t v4s;
#pragma unroll
            for (int qq = 0; qq < 4; qq++) {
                const uint32_t a = quad_bcast(offA, qq), b = quad_bcast(offB, qq);
                const bool tk = quad_bcast((uint32_t)take, qq) != 0;
                const v4s pa = {hold[qq][0], hold[qq][1], hold[qq][2], hold[qq][3]}, pb = {v[qq][0], v[qq][1], v[qq][2], v[qq][3]};
                __builtin_amdgcn_raw_buffer_store_b128(pa, orsrc, a + 16u * part, 0, 2);
                __builtin_amdgcn_raw_buffer_store_b128(pb, orsrc, b + 16u * part, 0, 2);
#pragma unroll
                for (int d = 0; d < 4; d++) hold[qq][d] = tk ? v[qq][d] : hold[qq][d];
            }
            held = take;
            hold_off = take ? O : hold_off;
        } else
        if constexpr (CAD) {                                      // four unconditional buffer stores: a lane of the quad without a full line asks for a dropped offset
#ifdef ABL_CAD_NO_STORE
            const int moff = (int)kDrop;                          // ablation: every line dropped
#else
            const int moff = (int)(full_out ? (uint32_t)((uint64_t)(uintptr_t)op - out_base) : kDrop);
#endif
#pragma unroll
            for (int qq = 0; qq < 4; qq++) {
                const uint32_t dof = (uint32_t)(qq == 0 ? __builtin_amdgcn_mov_dpp(moff, 0x00, 0xf, 0xf, true) : qq == 1 ? __builtin_amdgcn_mov_dpp(moff, 0x55, 0xf, 0xf, true)
                                              : qq == 2 ? __builtin_amdgcn_mov_dpp(moff, 0xAA, 0xf, 0xf, true) : __builtin_amdgcn_mov_dpp(moff, 0xFF, 0xf, 0xf, true));
                typedef __attribute__((__vector_size__(4 * sizeof(uint32_t)))) uint32_t v4s;
                const v4s piece = {v[qq][0], v[qq][1], v[qq][2], v[qq][3]};
                __builtin_amdgcn_raw_buffer_store_b128(piece, orsrc, dof + 16u * part, 0, 2);   // aux 2 = nt: streamed out once
            }
        } else {
        const uint64_t mine = full_out ? (uint64_t)(uintptr_t)op : 0ull;
        const int mlo = (int)(uint32_t)mine, mhi = (int)(uint32_t)(mine >> 32);
#pragma unroll
        for (int qq = 0; qq < 4; qq++) {
            uint32_t dlo, dhi;
            if (qq == 0) { dlo = (uint32_t)__builtin_amdgcn_mov_dpp(mlo, 0x00, 0xf, 0xf, true); dhi = (uint32_t)__builtin_amdgcn_mov_dpp(mhi, 0x00, 0xf, 0xf, true); }
            else if (qq == 1) { dlo = (uint32_t)__builtin_amdgcn_mov_dpp(mlo, 0x55, 0xf, 0xf, true); dhi = (uint32_t)__builtin_amdgcn_mov_dpp(mhi, 0x55, 0xf, 0xf, true); }
            else if (qq == 2) { dlo = (uint32_t)__builtin_amdgcn_mov_dpp(mlo, 0xAA, 0xf, 0xf, true); dhi = (uint32_t)__builtin_amdgcn_mov_dpp(mhi, 0xAA, 0xf, 0xf, true); }
            else { dlo = (uint32_t)__builtin_amdgcn_mov_dpp(mlo, 0xFF, 0xf, 0xf, true); dhi = (uint32_t)__builtin_amdgcn_mov_dpp(mhi, 0xFF, 0xf, 0xf, true); }
            const uint64_t da = ((uint64_t)dhi << 32) | dlo;
#ifdef ABL_NO_STORE
            if (da == 1) {
#else
            if (da) {
#endif
                v4u piece = {v[qq][0], v[qq][1], v[qq][2], v[qq][3]};
                __builtin_nontemporal_store(piece, (v4u_a1*)(uintptr_t)(da + 16u * part));   // streamed out once
            }
        }
        }
        if (!full_out && cnt) {
#pragma unroll
            for (int sN = 0; sN < 16; sN++) {
                const uint32_t done = 4u * sN;
                if (cnt >= done + 4) *(u32_a1*)(op + done) = wb[sN];
                else if (cnt > done) for (uint32_t k = 0; k < cnt - done; k++) op[done + k] = (uint8_t)(wb[sN] >> (8 * k));
            }
        }
        op += cnt;
        left -= cnt;
    }
    };
    if (shared) run(std::true_type{}); else run(std::false_type{});
    if constexpr (QW && HUF0_QW_PAIR) {                           // a lower half still waiting when the wave's last round ended
        const uint32_t offA = held ? hold_off : kDrop;
        typedef __attribute__((__vector_size__(4 * sizeof(uint32_t)))) uint32_t v4s;
#pragma unroll
        for (int qq = 0; qq < 4; qq++) {
            const uint32_t a = quad_bcast(offA, qq);
            const v4s pa = {hold[qq][0], hold[qq][1], hold[qq][2], hold[qq][3]};
            __builtin_amdgcn_raw_buffer_store_b128(pa, orsrc, a + 16u * part, 0, 2);
        }
    }
    if (streaming && P != 0) bad = true;                          // every stream ends exactly (BIT_endOfDStream)
    const bool any_bad = __builtin_amdgcn_mov_dpp((int)bad, 0x00, 0xf, 0xf, true) | __builtin_amdgcn_mov_dpp((int)bad, 0x55, 0xf, 0xf, true) |
                         __builtin_amdgcn_mov_dpp((int)bad, 0xAA, 0xf, 0xf, true) | __builtin_amdgcn_mov_dpp((int)bad, 0xFF, 0xf, 0xf, true);
    if (exists && j == 0 && rets) {
        if (mode == 1 || mode == 2) ret = (int64_t)dsize;
        else if (mode == 3 && ret == 0) ret = any_bad ? kCorrupt : (int64_t)dsize;
        rets[chunk] = ret;
    }
}

template <bool SO, int WG = 1, int PLOG = 4, bool CAD = false, int NS = 3, bool UA = false, bool QW = false>
__global__ void __launch_bounds__(64 * WG) __attribute__((amdgpu_waves_per_eu(SO ? (CAD && PLOG == 6 ? (NS == 2 ? 4 : (QW ? HUF0_QW_WAVES : 3)) : HUF0_SO_WAVES) : HUF0_G_WAVES)))   // (64-byte pieces: the LDS allows 10 waves a CU)
huf0_stream_kernel(const uint8_t* __restrict__ blocks, const uint64_t* __restrict__ boffs, uint64_t nchunks, uint8_t* __restrict__ out,
                   const uint64_t* __restrict__ ooffs, int64_t* __restrict__ rets, const uint8_t* __restrict__ desc, const uint8_t* __restrict__ share)
{
    huf0_stream_body<SO, WG, PLOG, CAD, NS, UA, QW>(blocks, boffs, nchunks, out, ooffs, rets, desc, share);
}

// small batches: both single-wave forms in ONE launch, a wave takes the one its segment's share flag names (where a launch is 2 - 3 %
// of the job, a second one in which every wave reads its flag and leaves is not free)
// (skip_shared: the one-tree segments have been taken by huf0_sync_kernel)
__global__ void __launch_bounds__(64) huf0_stream_small_kernel(const uint8_t* __restrict__ blocks, const uint64_t* __restrict__ boffs, uint64_t nchunks,
                                                               uint8_t* __restrict__ out, const uint64_t* __restrict__ ooffs, int64_t* __restrict__ rets,
                                                               const uint8_t* __restrict__ desc, const uint8_t* __restrict__ share, int skip_shared)
{
    if (share[(uint64_t)blockIdx.x * 16 >> 6] != 0) {
        if (skip_shared) return;
        huf0_stream_body<true, 1, HUF0_SMALL_PLOG, HUF0_SMALL_CAD != 0, 3, true>(blocks, boffs, nchunks, out, ooffs, rets, desc, share);
    }
    else huf0_stream_body<false, 1, HUF0_G_PLOG, HUF0_G_CAD != 0, 3, true>(blocks, boffs, nchunks, out, ooffs, rets, desc, share);
}

#include "huf0_sync.h"

std::string g_err0;

}  // namespace

extern "C" {

// descriptors | follow flags (a multiple of 64) | share flags (one per segment)
size_t sprintz_mi355x_huf0_decode_tmp_bytes(uint64_t nchunks)
{
    return (size_t)nchunks * kDescStride + 256 + (((size_t)nchunks + 63) & ~(size_t)63) + (size_t)((nchunks + 63) / 64) + 64;
}

int sprintz_mi355x_huf0_decompress_batch_ws(const void* d_blocks, const uint64_t* d_block_offsets, uint64_t nchunks, void* d_out,
                                            const uint64_t* d_out_offsets, int64_t* d_rets, void* d_tmp, void* hip_stream)
{
    return sprintz_mi355x_huf0_decompress_batch_hint(d_blocks, d_block_offsets, nchunks, d_out, d_out_offsets, d_rets, d_tmp, 0, hip_stream);
}

int sprintz_mi355x_huf0_decompress_batch_hint(const void* d_blocks, const uint64_t* d_block_offsets, uint64_t nchunks, void* d_out,
                                              const uint64_t* d_out_offsets, int64_t* d_rets, void* d_tmp, uint32_t max_block_bytes, void* hip_stream)
{
    if (!d_blocks || !d_block_offsets || !d_out || !d_out_offsets || ((uintptr_t)d_blocks & 15) || (nchunks && (!d_tmp || ((uintptr_t)d_tmp & 15))))
        return sprintz::set_error(SPRINTZ_E_INVALID, "Huff0 stage: invalid argument (null pointer, alignment or size)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return sprintz::set_error(SPRINTZ_E_NO_DEVICE, "Huff0 stage: no usable HIP device (there is no CPU fallback)");
    if (nchunks == 0) return 0;
    const uint64_t grid1 = (nchunks + 63) / 64, grid2 = (nchunks + 15) / 16, nleaders = grid1;
    if (grid2 > 0x7fffffffull) return sprintz::set_error(SPRINTZ_E_INVALID, "Huff0 stage: invalid argument (null pointer, alignment or size)");
    hipStream_t st = (hipStream_t)hip_stream;
    uint8_t* const desc = (uint8_t*)d_tmp;
    uint8_t* const follow = desc + (((size_t)nchunks * kDescStride + 255) & ~(size_t)255);
    uint8_t* const share = follow + ((nchunks + 63) & ~(uint64_t)63);
    const uint8_t* const blk = (const uint8_t*)d_blocks;
    // Up to HUF0_WAVE_LEADERS segments: ONE kernel, a wave per segment, does the follow test, the leader's tree, the followers'
    // copies and the share flag (46 us instead of 112 for the leader's tree at 157 .. 1 250 leaders, and three launches less).
    // Above: the follow pass, a LANE per leader, the share pass, the copy pass, then huf0_tree_kernel<2> for every chunk that is neither
    // leader nor follower (the wave kernel does those itself).  (Round 4 drew the line at 4 096 segments: the lane-per-leader parse
    // alone is 0.13 ms at 12 500 leaders where the wave kernel takes 0.22 -- but the wave kernel's 0.22 is the follow pass's 0.08 and the
    // other passes as well: at 800 000 chunks the whole stage 2.23 against 2.28 - 2.31 ms with it, round 5.)
    if (nleaders <= HUF0_WAVE_LEADERS) {
        if (nleaders <= 1024) {
            hipLaunchKernelGGL(huf0_tree_wave_kernel<true>, dim3((unsigned)nleaders), dim3(64), 0, st, blk, d_block_offsets, d_out_offsets, nchunks, desc, follow, share);
        } else {
            hipLaunchKernelGGL(huf0_tree_wave_kernel<false>, dim3((unsigned)nleaders), dim3(64), 0, st, blk, d_block_offsets, d_out_offsets, nchunks, desc, follow, share);
            hipLaunchKernelGGL(huf0_tree_kernel<2>, dim3((unsigned)grid1), dim3(64), 0, st, blk, d_block_offsets, d_out_offsets, nchunks, desc,
                               (const uint8_t*)follow);
        }
    } else {
        hipLaunchKernelGGL(huf0_follow_kernel, dim3((unsigned)((nchunks + 255) / 256)), dim3(256), 0, st, blk, d_block_offsets, d_out_offsets, nchunks, follow);
        hipLaunchKernelGGL(huf0_tree_kernel<1>, dim3((unsigned)((nleaders + 63) / 64)), dim3(64), 0, st, blk, d_block_offsets, d_out_offsets, nchunks, desc,
                           (const uint8_t*)follow);
        hipLaunchKernelGGL(huf0_share_kernel, dim3((unsigned)((nleaders + 255) / 256)), dim3(256), 0, st, (const uint8_t*)desc, (const uint8_t*)follow, nchunks, share);
        hipLaunchKernelGGL(huf0_copy_kernel, dim3((unsigned)((nchunks * 20 + 255) / 256)), dim3(256), 0, st, desc, (const uint8_t*)follow, nchunks, (const uint8_t*)share);
        hipLaunchKernelGGL(huf0_tree_kernel<2>, dim3((unsigned)grid1), dim3(64), 0, st, blk, d_block_offsets, d_out_offsets, nchunks, desc,
                           (const uint8_t*)follow);
    }
    // the one-table kernel: bandwidth-sized batches as workgroups of HUF0_BIG_WG waves with 2^HUF0_BIG_PLOG-byte stream pieces (built: 2 waves, 64 bytes),
    // then the per-chunk-table kernel for the segments that are not its; small batches: both wave by wave in one launch
    if (nchunks >= (uint64_t)sprintz::huf0_big_batch().load(std::memory_order_relaxed)) {
        hipLaunchKernelGGL((huf0_stream_kernel<true, HUF0_BIG_WG, HUF0_BIG_PLOG, HUF0_CADENCED != 0, HUF0_BIG_NS, HUF0_BIG_UA != 0, HUF0_BIG_QW != 0>), dim3((unsigned)((nchunks + 16 * HUF0_BIG_WG - 1) / (16 * HUF0_BIG_WG))),
                           dim3(64 * HUF0_BIG_WG), 0, st, blk, d_block_offsets, nchunks,
                           (uint8_t*)d_out, d_out_offsets, d_rets, (const uint8_t*)desc, (const uint8_t*)share);
        hipLaunchKernelGGL((huf0_stream_kernel<false, 1, HUF0_G_PLOG, HUF0_G_CAD != 0>), dim3((unsigned)grid2), dim3(64), 0, st, blk, d_block_offsets, nchunks,
                           (uint8_t*)d_out, d_out_offsets, d_rets, (const uint8_t*)desc, (const uint8_t*)share);
    } else if (nchunks <= (uint64_t)sprintz::huf0_sync_chunks().load(std::memory_order_relaxed)) {
        // small batches: a wave per chunk, sixteen self-synchronising decoders per stream (huf0_sync.h), one launch for both kinds of segment.
        // The block image a wave keeps in LDS is sized by the caller's hint (the largest block of the batch; a block above the image is read
        // from global memory by the same code)
        constexpr int kWpb = 4;
        const uint32_t want = max_block_bytes ? max_block_bytes : 4096u;
        const uint32_t img = ((want < 16384u ? want : 16384u) + 16u + 63u) & ~63u;
        const size_t lds = sync_lds_bytes(kWpb, img);
        // (chunks a wave takes one after the other: measured 1 / 2 / 4 -> 43.8 / 60.0 / 91.5 us at 625 chunks, 106 / 99.5 / 119 at 10 000 -- tools/huf0_sync_ab.sh,
        //  round 5; the environment knob that chose it had no test and is gone: one chunk a wave)
        const uint32_t cpw = 1u;
        const uint64_t per_wg = (uint64_t)kWpb * cpw;
        if (sprintz::launch_with_lds(huf0_sync_kernel<kWpb>, (unsigned)((nchunks + per_wg - 1) / per_wg), 64u * kWpb, lds, st, blk, d_block_offsets, nchunks,
                                     (uint8_t*)d_out, d_out_offsets, d_rets, (const uint8_t*)desc, (const uint8_t*)share, img, cpw) != hipSuccess)
            return sprintz::set_error(SPRINTZ_E_HIP, "Huff0 stage: the self-synchronising stream kernel's launch failed");
    } else {
        hipLaunchKernelGGL(huf0_stream_small_kernel, dim3((unsigned)grid2), dim3(64), 0, st, blk, d_block_offsets, nchunks,
                           (uint8_t*)d_out, d_out_offsets, d_rets, (const uint8_t*)desc, (const uint8_t*)share, 0);
    }
    return hipGetLastError() == hipSuccess ? 0 : sprintz::set_error(SPRINTZ_E_HIP, "Huff0 stage: a HIP call or kernel launch failed");
}

#ifdef HUF0_TREE_TIMING
int sprintz_mi355x_dbg_tree_stamps(uint64_t* out16) { return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_tree_ts), 16 * sizeof(uint64_t)); }
#endif

// the ABI-1 form without a workspace argument: stream-ordered allocation of the descriptors
int sprintz_mi355x_huf0_decompress_batch(const void* d_blocks, const uint64_t* d_block_offsets, uint64_t nchunks, void* d_out,
                                         const uint64_t* d_out_offsets, int64_t* d_rets, void* hip_stream)
{
    if (!d_blocks || !d_block_offsets || !d_out || !d_out_offsets || ((uintptr_t)d_blocks & 15))
        return sprintz::set_error(SPRINTZ_E_INVALID, "Huff0 stage: invalid argument (null pointer, alignment or size)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return sprintz::set_error(SPRINTZ_E_NO_DEVICE, "Huff0 stage: no usable HIP device (there is no CPU fallback)");
    if (nchunks == 0) return 0;
    void* tmp = nullptr;
    if (hipMallocAsync(&tmp, sprintz_mi355x_huf0_decode_tmp_bytes(nchunks), (hipStream_t)hip_stream) != hipSuccess)
        return sprintz::set_error(SPRINTZ_E_HIP, "Huff0 stage: hipMallocAsync of the descriptor workspace failed");
    const int rc = sprintz_mi355x_huf0_decompress_batch_ws(d_blocks, d_block_offsets, nchunks, d_out, d_out_offsets, d_rets, tmp, hip_stream);
    (void)hipFreeAsync(tmp, (hipStream_t)hip_stream);
    return rc;
}

}  // extern "C"

// huf0_sync.h -- the Huff0 stream stage for batches that leave the chip empty: one WAVE per chunk, SIXTEEN decoders per stream.
// Included by huf0.hip (inside its anonymous namespace, behind the descriptor constants); same inputs, same outputs, same return
// values as huf0_stream_kernel.
//
// A Huff0 stream is ONE serial chain of look-ups (~900 for a 10 KB chunk of bit-packed Sprintz output: ~70 us however few chunks
// there are -- a lone wave gets a dependent LDS round trip plus three vector instructions done per ~75 ns).  But a prefix code
// SELF-SYNCHRONISES: a decoder started at an arbitrary bit decodes garbage only until its cursor happens to land on a true code
// boundary, and on these streams that is soon -- tools/huf0_resync_stats.py, the bench's own chunks: half of all wrong starts are
// right after 6 symbols, 90 % after 19, 99 % after 39 (279 bits), none later than 512 symbols.  So (Weissenberger & Schmidt's
// scheme, ICPP 2018, cut down to what a wave can do in registers):
//   * a stream's bits are cut into 16 equal SPANS; lane k of the stream's 16 lanes owns span k (the stream is read from its end:
//     span 0 is the top);
//   * COUNT pass from a lane's current start to the first code boundary at or below its span's end -> (end cursor E, symbols n).
//     Iteration 0 starts every lane at its span's top -- a guess; then every lane restarts where its upper neighbour ended
//     (lane 0: the stream's end mark, which is a true boundary) until no lane's E moves: by induction from lane 0 every start is then a
//     true boundary and every n the true count.  A lane's guess is wrong only if its own path had not synchronised within its span
//     (~400 bits here): two iterations for most waves, a third for a few;
//   * a prefix sum of n over the stream's lanes places every lane's symbols; EMIT pass: decode again from the true start, four
//     symbols -> one dword -> one store.
// Three passes over a span of ~60 symbols instead of one over 900: the chain is ~5x shorter, the look-ups 3x as many -- which is why
// this is the SMALL-batch form (below about one resident wave a SIMD per chunk the chip has the issue slots to spare; a large batch is
// bound by instructions and keeps huf0_stream_kernel).  The whole block sits in LDS first (ONE coalesced read), so the decoders' windows
// are LDS reads; a block larger than the launch's image (or than LDS) is read from global memory by the same code (slow, correct).
// The decode table is the full 2^tableLog one (length | symbol << 8), ONE per workgroup (a table per wave would be 8 KB of LDS a chunk:
// a third of the resident waves).  Where a segment's chunks share a tree (share[] != 0: our writer, any writer that repeats a
// description) the workgroup's waves decode their chunks side by side behind one table build; where they do not (libzstd's blocks: a
// private tree per block) the workgroup takes its chunks one after the other, the table rebuilt in between -- the same launch, the same
// code, no 8-bit prefix tables.  A wave takes `cpw` consecutive rounds of chunks (large "small" batches: all waves resident at once).
// Pass B of the count also KEEPS what it decodes (up to 96 + 8 symbols a lane in registers): when no lane's start moves afterwards --
// the usual case -- the symbols are stored from there and the emit pass is skipped.

constexpr int kSyncTab = 2 << 12;                  // bytes of the decode table (table log <= 12)
constexpr int kSyncApron = 32;                     // bytes in front of a wave's block image
constexpr int kSyncKeep = 24;                      // dwords of decoded symbols a lane keeps from the counting pass

__host__ __device__ constexpr uint32_t sync_wave_bytes(uint32_t img) { return kSyncApron + img + 16; }   // (+ 16: a window's third dword may lie past the image)
__host__ __device__ constexpr uint32_t sync_lds_bytes(int wpb, uint32_t img) { return kSyncTab + kDescStride + (uint32_t)wpb * sync_wave_bytes(img); }

// One wave, one chunk: everything behind the table build.  `mine`: this wave decodes (wave-uniform); s_tab holds the chunk's table,
// s_desc its descriptor.  Called by all waves of the workgroup (it contains no barrier).
struct SyncChunk {
    const uint8_t* src;
    uint8_t* dst;
    uint64_t csize, dsize;
    bool exists;
};

template <int WPB>
__global__ void __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(6, 8))) huf0_sync_kernel(const uint8_t* __restrict__ blocks, const uint64_t* __restrict__ boffs, uint64_t nchunks,
                                                             uint8_t* __restrict__ out, const uint64_t* __restrict__ ooffs, int64_t* __restrict__ rets,
                                                             const uint8_t* __restrict__ desc, const uint8_t* __restrict__ share, uint32_t img_bytes, uint32_t cpw)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_sync[];
    typedef __attribute__((address_space(3))) const uint16_t lds_u16;
    typedef __attribute__((address_space(3))) const uint32_t lds_u32c;
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    typedef v4u __attribute__((aligned(16), may_alias)) v4u_a16;
    typedef uint64_t __attribute__((aligned(1), may_alias)) u64_a1;
    const int t = threadIdx.x, wv = t >> 6, lane = t & 63, j = lane >> 4, k = lane & 15;
    uint8_t* const s_tab = s_sync;
    uint8_t* const s_desc = s_sync + kSyncTab;
    uint8_t* const s_img = s_desc + kDescStride + (uint32_t)wv * sync_wave_bytes(img_bytes) + kSyncApron;     // 16-byte aligned
    const uint32_t ft = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)s_tab;
    const uint32_t im = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)s_img;
    auto uni64 = [](uint64_t v) {
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    };
    // the descriptor of chunk c (sorted symbols | table words) -> LDS, by the first 20 lanes of the workgroup
    auto stage_desc = [&](uint64_t c) {
        if (t < 20) {
            v4u v = {0, 0, 0, 0};
            if (c < nchunks) v = *(const v4u_a16*)(desc + c * kDescStride + 16u * (uint32_t)t);
            uint32_t* const d = (uint32_t*)(s_desc + 16u * (uint32_t)t);
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
    };
    // the decode table of the staged descriptor: entry idx = length | symbol << 8 for the code the tl look-ahead bits idx start with
    auto build_table = [&]() {
        const uint32_t* const tab = (const uint32_t*)(s_desc + 256);
        const uint32_t tab0 = tab[0], tl = tab0 >> 16;
        if ((tab0 & 0xffffu) == 0u || tl == 0u || tl > 12u) return;
        const uint32_t size = 1u << tl, per = size >= 64u * WPB ? size / (64u * WPB) : 1u;
        uint16_t* const tb = (uint16_t*)s_tab;
        uint32_t w = 1, e = tab[1], nxt = tab[2] & 0xffffu;
        for (uint32_t i = 0; i < per; i++) {
            const uint32_t idx0 = (uint32_t)t * per + i;
            if (idx0 >= size) break;
            while (w < 12u && idx0 >= nxt) { w++; e = tab[w]; nxt = w < 12u ? (tab[w + 1] & 0xffffu) : 0xffffffffu; }
            const uint32_t pos = (e >> 16) + ((idx0 - (e & 0xffffu)) >> (w - 1u));
            const uint32_t len = tl + 1u - w;
            tb[idx0] = (uint16_t)((len >= 1u && len <= 12u ? len : 1u) | ((uint32_t)s_desc[pos & 0xffu] << 8));   // (a length is never 0: every pass advances)
        }
    };
    // a chunk's ranges, wave-uniform
    auto ranges = [&](uint64_t c) -> SyncChunk {
        SyncChunk r;
        r.exists = c < nchunks;
        const uint64_t b0 = r.exists ? uni64(boffs[c]) : 0, b1 = r.exists ? uni64(boffs[c + 1]) : 0;
        const uint64_t o0 = r.exists ? uni64(ooffs[c]) : 0, o1 = r.exists ? uni64(ooffs[c + 1]) : 0;
        r.src = blocks + b0; r.dst = out + o0;
        r.csize = b1 - b0; r.dsize = o1 - o0;
        if (r.exists && (b1 < b0 || o1 < o0)) { r.csize = 1; r.dsize = 0; }      // (reads as damaged below)
        return r;
    };
    // HUF_decompress's conventions (huf_decompress.c): 0 nothing / damaged, 1 stored, 2 one repeated byte, 3 coded
    auto mode_of = [&](const SyncChunk& r, int64_t& ret) -> int {
        ret = 0;
        if (!r.exists) return 0;
        if (r.dsize == 0) { ret = r.csize == 0 ? 0 : kCorrupt; return 0; }
        if (r.csize == 0 || r.csize > r.dsize) { ret = kCorrupt; return 0; }
        return r.csize == r.dsize ? 1 : r.csize == 1 ? 2 : 3;
    };
    // the block -> this wave's LDS image in one coalesced read: 16-byte pieces from the 16-byte line it starts in (image byte i = block byte i - b_al)
    auto load_image = [&](const SyncChunk& r, int mode) -> bool {
        const uint32_t b_al = (uint32_t)((uintptr_t)r.src & 15u);
        const bool in_lds = mode == 3 && r.csize < (1u << 27) && (uint32_t)r.csize + b_al <= img_bytes;
        if (in_lds) {
            const uint8_t* const g = r.src - b_al;
            const uint32_t span_bytes = (uint32_t)r.csize + b_al;
            for (uint32_t i = (uint32_t)lane * 16u; i < span_bytes; i += 1024u) {
                const v4u v = *(const v4u_a16*)(g + i);           // (the last piece may reach past the block by < 16 bytes: inside the buffer's last line, or the caller's slack)
                uint32_t* const d = (uint32_t*)(s_img + i);
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        }
        return in_lds;
    };

    // ---- one wave decodes one chunk (table and descriptor in LDS, image loaded if in_lds)
    auto decode = [&](uint64_t chunk, const SyncChunk& r, int mode, int64_t ret, bool in_lds) {
        const uint8_t* const src = r.src;
        uint8_t* const dst = r.dst;
        const uint64_t csize = r.csize, dsize = r.dsize;
        if (mode == 1) {                                          // stored: the wave copies
            for (uint64_t i = (uint64_t)lane * 4; i + 4 <= dsize; i += 256) *(u32_a1*)(dst + i) = *(const u32_a1*)(src + i);
            if (lane < (int)(dsize & 3)) dst[(dsize & ~(uint64_t)3) + (uint64_t)lane] = src[(dsize & ~(uint64_t)3) + (uint64_t)lane];
        }
        if (mode == 2) {
            const uint32_t v = src[0] * 0x01010101u;
            for (uint64_t i = (uint64_t)lane * 4; i + 4 <= dsize; i += 256) *(u32_a1*)(dst + i) = v;
            if (lane < (int)(dsize & 3)) dst[(dsize & ~(uint64_t)3) + (uint64_t)lane] = (uint8_t)v;
        }
        const uint32_t* const tab = (const uint32_t*)(s_desc + 256);
        const uint32_t tab0 = mode == 3 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)tab[0]) : 0u;
        const uint32_t hl = tab0 & 0xffffu, tl = tab0 >> 16;
        if (mode == 3 && (hl == 0u || tl == 0u || tl > 12u || hl >= csize)) { ret = kCorrupt; mode = 0; }
        const bool coded = mode == 3;                             // (wave-uniform)
        const uint32_t b_al = (uint32_t)((uintptr_t)src & 15u);

        // lane (j, k): sub-sequence k of stream j (HUF_decompress4X1_usingDTable_internal's four streams)
        bool bad = false;
        uint32_t sbyte = 0;                                       // the stream's first byte, as an offset into the image (in_lds) / from src - b_al
        int32_t Pmax = 0;
        uint32_t left = 0;                                        // symbols of stream j
        uint8_t* op = dst;
        const uint8_t* const gimg = src - b_al;                   // what image byte 0 is in global memory
        auto img_byte = [&](uint32_t i) -> uint32_t { return in_lds ? s_img[i] : gimg[i]; };
        if (coded) {
            const uint64_t n = csize - hl;
            const uint32_t jt = b_al + hl;                        // the jump table
            if (n < 10 || csize >= (1u << 27)) bad = true;        // (bit cursors are 32 bits; a Huff0 block is at most 128 KB)
            uint32_t l0 = 0, l1 = 0, l2 = 0, l3 = 0;
            if (!bad) {
                l0 = img_byte(jt) | (img_byte(jt + 1) << 8);
                l1 = img_byte(jt + 2) | (img_byte(jt + 3) << 8);
                l2 = img_byte(jt + 4) | (img_byte(jt + 5) << 8);
                if (6ull + l0 + l1 + l2 > n) bad = true;
                else l3 = (uint32_t)(n - 6 - l0 - l1 - l2);
            }
            if (!bad) {
                const uint64_t seg = (dsize + 3) / 4;
                const uint32_t so = 6u + (j > 0 ? l0 : 0u) + (j > 1 ? l1 : 0u) + (j > 2 ? l2 : 0u);
                const uint32_t slen = j == 0 ? l0 : j == 1 ? l1 : j == 2 ? l2 : l3;
                uint64_t w0 = seg * (uint64_t)j;
                w0 = w0 < dsize ? w0 : dsize;
                const uint64_t w1 = j == 3 ? dsize : (w0 + seg < dsize ? w0 + seg : dsize);
                sbyte = jt + so;
                const uint32_t lastb = slen >= 1 ? img_byte(sbyte + slen - 1) : 0u;
                if (slen < 1 || lastb == 0) bad = true;
                else {
                    Pmax = 8 * (int32_t)(slen - 1) + highbit(lastb);
                    op = dst + w0;
                    left = (uint32_t)(w1 - w0);
                }
            }
        }
        // a damaged stream damages the chunk: nothing of it is decoded
        const bool chunk_bad0 = __ballot(coded && bad) != 0;
        const bool active = coded && !chunk_bad0;
        const uint32_t span = ((uint32_t)Pmax + 15u) >> 4;
        const int32_t Gk = (int32_t)Pmax - (int32_t)(span * (uint32_t)k) > 0 ? (int32_t)Pmax - (int32_t)(span * (uint32_t)k) : 0;
        const int32_t Gn = (k == 15 || (int32_t)Pmax - (int32_t)(span * (uint32_t)(k + 1)) < 0) ? 0 : (int32_t)Pmax - (int32_t)(span * (uint32_t)(k + 1));
        const uint32_t look_shift = 32u - (tl ? tl : 1u);
        const int32_t fast_margin = 3 * (int32_t)tl;              // above Gn + this, four symbols all START above Gn
        // the 8 bytes that end at the cursor's byte, the cursor's bit on top (P >= 1).  Bytes below the stream's first are whatever precedes it
        // (the jump table, the previous stream): a valid stream never consumes them -- they only fill up the last look-aheads
        auto window = [&](int32_t P) -> uint64_t {
            const uint32_t pm1 = (uint32_t)P - 1u;
            const uint32_t o = sbyte + (pm1 >> 3) - 7u;          // (sbyte >= 7: a header byte and the jump table lie in front)
            uint64_t w;
            if (in_lds) {
                const uint32_t a = im + (o & ~3u);
                const uint32_t d0 = *(lds_u32c*)(uintptr_t)a, d1 = *(lds_u32c*)(uintptr_t)(a + 4u), d2 = *(lds_u32c*)(uintptr_t)(a + 8u);
                const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, o & 3u), hi = __builtin_amdgcn_alignbyte(d2, d1, o & 3u);
                w = ((uint64_t)hi << 32) | lo;
            } else {
                w = *(const u64_a1*)(gimg + o);
            }
            return w << (7u - (pm1 & 7u));
        };
        auto lookup = [&](uint64_t win) -> uint32_t { return *(lds_u16*)(uintptr_t)(ft + (((uint32_t)(win >> 32) >> look_shift) << 1)); };

        // COUNT: from `start` down to the first boundary at or below Gn.  Wave-uniform loops with a guard: every symbol takes >= 1 bit.
        // KEEP: the symbols stay in registers (kb: whole steps of four; kt: the masked steps' last few) -- `over` if they do not fit
        const uint32_t guard_max = (uint32_t)(csize >> 1) + 80u; // (a span is at most 8 * csize / 16 bits, wave-uniform)
        uint32_t kb[kSyncKeep], kt[2];
        uint32_t nfast = 0;                                       // whole steps this lane kept
        bool over = false;
        auto count_pass = [&](int32_t start, bool on, int32_t& E, uint32_t& n, auto KEEP) {
            constexpr bool kKeep = decltype(KEEP)::value;
            int32_t P = start;
            uint32_t cnt = 0;
            if constexpr (kKeep) {
#pragma unroll
                for (int g = 0; g < kSyncKeep; g++) {
                    const bool go = on && P - Gn > fast_margin;
                    if (__ballot(go) == 0) break;
                    if (go) {
                        uint64_t win = window(P);
                        const uint32_t e0 = lookup(win); win <<= e0 & 63u;
                        const uint32_t e1 = lookup(win); win <<= e1 & 63u;
                        const uint32_t e2 = lookup(win); win <<= e2 & 63u;
                        const uint32_t e3 = lookup(win);
                        P -= (int32_t)((e0 + e1 + e2 + e3) & 0xffu);
                        const uint32_t w01 = __builtin_amdgcn_perm(e1, e0, 0x0c0c0501u), w23 = __builtin_amdgcn_perm(e3, e2, 0x0c0c0501u);
                        kb[g] = __builtin_amdgcn_perm(w23, w01, 0x05040100u);
                        cnt += 4;
                    }
                }
                if (__ballot(on && P - Gn > fast_margin) != 0) over = true;    // (wave-uniform) more whole steps than registers
            }
            for (uint32_t g = 0; g < guard_max; g++) {            // four symbols a trip while all four start above Gn
                const bool go = on && P - Gn > fast_margin;
                if (__ballot(go) == 0) break;
                if (go) {
                    uint64_t win = window(P);
                    const uint32_t e0 = lookup(win); win <<= e0 & 63u;
                    const uint32_t e1 = lookup(win); win <<= e1 & 63u;
                    const uint32_t e2 = lookup(win); win <<= e2 & 63u;
                    const uint32_t e3 = lookup(win);
                    P -= (int32_t)((e0 + e1 + e2 + e3) & 0xffu);
                    cnt += 4;
                }
            }
            if constexpr (kKeep) { if (on) nfast = cnt >> 2; }
#pragma unroll 1
            for (uint32_t g = 0; g < 16u; g++) {                  // the last few: a symbol counts if it STARTS above Gn
                const bool go = on && P > Gn;
                if (__ballot(go) == 0) break;
                if (kKeep && g >= 2u) over = true;
                if (go) {
                    uint64_t win = window(P);
                    uint32_t word = 0;
#pragma unroll
                    for (int s = 0; s < 4; s++) {
                        const uint32_t e = lookup(win);
                        const bool take = P > Gn;
                        const uint32_t nb = take ? (e & 0xffu) : 0u;
                        win <<= nb;
                        P -= (int32_t)nb;
                        cnt += take ? 1u : 0u;
                        word |= (take ? (e >> 8) & 0xffu : 0u) << (8 * s);
                    }
                    if constexpr (kKeep) { kt[0] = g == 0u ? word : kt[0]; kt[1] = g == 1u ? word : kt[1]; }
                }
            }
            E = P;
            n = cnt;
        };
        int32_t E = 0, T = Gk;
        uint32_t n = 0;
        kt[0] = kt[1] = 0;
        count_pass(T, active, E, n, std::false_type{});           // pass A: every lane from its span's top (a guess)
        bool settled = false;
        for (int it = 0; it < 17; it++) {
            // the upper neighbour's end is this lane's start (row_shr:1 inside the stream's 16 lanes; lane 0: the end mark)
            const int32_t up = __builtin_amdgcn_update_dpp(0, E, 0x111, 0xf, 0xf, false);
            const int32_t Tn = k == 0 ? Pmax : up;
            const bool redo = active && (it == 0 || Tn != T);
            if (it > 0 && __ballot(redo) == 0) { settled = true; break; }
            T = Tn;
            int32_t E2 = E;
            uint32_t n2 = n;
            count_pass(T, redo, E2, n2, std::true_type{});        // pass B (and again for the lanes whose start moved)
            if (redo) { E = E2; n = n2; }
        }
        // (17 iterations always settle a valid stream: lane k is final after iteration k)
        bool sbad = active && !settled;
        // every stream ends exactly (BIT_endOfDStream) and holds exactly its share of the symbols
        uint32_t incl = n;                                        // inclusive scan over the stream's 16 lanes
        {
            uint32_t v;
            v = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xf, 0xf, false); incl += v;      // row_shr:1
            v = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xf, 0xf, false); incl += v;      // row_shr:2
            v = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xf, 0xf, false); incl += v;      // row_shr:4
            v = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xf, 0xf, false); incl += v;      // row_shr:8
        }
        if (active && k == 15 && (E != 0 || incl != left)) sbad = true;
        const bool chunk_bad = chunk_bad0 || __ballot(sbad) != 0;

        if (active && !chunk_bad) {
            uint8_t* p = op + (incl - n);
            if (!over) {
                // the symbols are in registers: whole steps, then the masked steps' bytes
#pragma unroll
                for (int g = 0; g < kSyncKeep; g++) {
                    if (__ballot((uint32_t)g < nfast) == 0) break;
                    if ((uint32_t)g < nfast) *(u32_a1*)(p + 4 * g) = kb[g];
                }
                uint8_t* q = p + 4u * nfast;
                const uint32_t rest = n - 4u * nfast;             // 0 .. 8
                const uint64_t tail = (uint64_t)kt[0] | ((uint64_t)kt[1] << 32);
                if (rest >= 4u) { *(u32_a1*)q = kt[0]; }
                for (uint32_t s = rest >= 4u ? 4u : 0u; s < rest; s++) q[s] = (uint8_t)(tail >> (8 * s));
            } else {
                // EMIT: from the true start, n symbols to op + (symbols of the lanes above)
                int32_t P = T;
                uint32_t cnt = n;
                for (uint32_t g = 0; g < guard_max; g++) {
                    const bool go = cnt >= 4u;
                    if (__ballot(go) == 0) break;
                    if (go) {
                        uint64_t win = window(P);
                        const uint32_t e0 = lookup(win); win <<= e0 & 63u;
                        const uint32_t e1 = lookup(win); win <<= e1 & 63u;
                        const uint32_t e2 = lookup(win); win <<= e2 & 63u;
                        const uint32_t e3 = lookup(win);
                        P -= (int32_t)((e0 + e1 + e2 + e3) & 0xffu);
                        const uint32_t w01 = __builtin_amdgcn_perm(e1, e0, 0x0c0c0501u), w23 = __builtin_amdgcn_perm(e3, e2, 0x0c0c0501u);
                        *(u32_a1*)p = __builtin_amdgcn_perm(w23, w01, 0x05040100u);
                        p += 4;
                        cnt -= 4;
                    }
                }
                if (cnt) {                                        // the lane's last 1 .. 3 symbols
                    uint64_t win = window(P);
                    for (uint32_t s = 0; s < cnt; s++) {
                        const uint32_t e = lookup(win);
                        win <<= e & 63u;
                        p[s] = (uint8_t)(e >> 8);
                    }
                }
            }
        }
        if (r.exists && lane == 0 && rets) {
            if (mode == 1 || mode == 2) ret = (int64_t)dsize;
            else if (coded) ret = chunk_bad ? kCorrupt : (int64_t)dsize;
            rets[chunk] = ret;
        }
    };

    // ---- the workgroup's chunks: cpw rounds of WPB consecutive chunks (a round lies inside one 64-chunk segment).  A round is ONE step
    // where the segment has one tree (every wave its own chunk, side by side behind one table) and WPB steps where every chunk has its
    // own (step i: the table of chunk i, wave i decodes it)
    const uint64_t first = (uint64_t)blockIdx.x * WPB * cpw;
    uint64_t table_of = ~0ull;                                    // the segment whose shared table is in LDS
    bool started = false;
    for (uint32_t it = 0; it < cpw * WPB; it++) {
        const uint32_t rd = it / WPB, step = it % WPB;
        const uint64_t c0 = first + (uint64_t)rd * WPB;
        if (c0 >= nchunks) break;                                 // (workgroup-uniform)
        const bool sh = share[c0 >> 6] != 0;
        if (sh && step) continue;
        const uint64_t tchunk = sh ? (c0 & ~(uint64_t)63) : c0 + step;      // whose descriptor the table is built from (one tree: the leader's -- large batches leave its followers without copies)
        if (tchunk >= nchunks) continue;
        const bool mine = sh || wv == (int)step;                  // (wave-uniform)
        const uint64_t chunk = sh ? c0 + (uint64_t)wv : tchunk;
        const SyncChunk r = ranges(mine ? chunk : nchunks);
        int64_t ret;
        const int mode = mode_of(r, ret);
        const bool rebuild = !(sh && table_of == (c0 >> 6));
        if (started) __syncthreads();                             // the last step's readers are done with the table and the descriptor
        started = true;
        if (rebuild) stage_desc(tchunk);
        const bool in_lds = load_image(r, mode);
        if (rebuild) { __syncthreads(); build_table(); }
        __syncthreads();
        table_of = sh ? (c0 >> 6) : ~0ull;
        decode(chunk, r, mode, ret, in_lds);
    }
}

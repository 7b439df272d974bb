// huf0_write.h -- writer of Huff0-format blocks (included by huf.hip inside its anonymous namespace:
// it reuses K1, huf_build_kernel, for the per-segment code lengths).  The blocks are specified by
// oracle/huf0_oracle.c's oracle_huf0_compress_batch (one code table per segment of 64 chunks,
// written into every chunk's block; Huff0's canonical code order; FSE-coded weights with table log 6
// or the 4-bit form) and are byte-exact with it; the library's HUF_decompress reads them.
//   H1 huf0_table_kernel   per segment: lengths -> tree description bytes + code values (one wave;
//                          the FSE state chain is searched 64 table entries at a time)
//   H2 huf0_size_kernel    per chunk: the four streams' byte counts, stored / repeated / coded
//   H3 huf0_encode_kernel  per chunk: description, jump table, four streams written last symbol
//                          first through a per-lane LDS ring that leaves in 64-byte units
#pragma once

// per-segment record (H1 -> H2, H3): hlen u32 @0 (0 = no table: the segment's chunks are stored) | tableLog u32 @4 |
// tree description, <= 160 bytes @8 | code values 256 x u16 @192 | code lengths 256 x u8 @704
constexpr int kRecBytes = 1024, kRecHdr = 8, kRecTab = 192;
#ifndef HUF0S_PAD
#define HUF0S_PAD 40000
#endif
constexpr unsigned kSizePassPad = HUF0S_PAD, kSizePassPadFrom = 4096;      // huf0_size_kernel: dynamic LDS claimed (unused) from this many workgroups on

struct BitW {                           // bytes into LDS, LSB first (bitstream.h BIT_CStream_t)
    uint8_t* p; uint64_t acc; int nbits; uint32_t n;
    __device__ void add(uint32_t v, int nb)
    {
        acc |= (uint64_t)v << nbits;
        nbits += nb;
        while (nbits >= 8) { p[n++] = (uint8_t)acc; acc >>= 8; nbits -= 8; }
    }
    __device__ uint32_t close(bool end_mark)
    {
        if (end_mark) add(1, 1);
        if (nbits > 0) { p[n++] = (uint8_t)acc; acc = 0; nbits = 0; }
        return n;
    }
};

// Round 5: the whole wave works.  (Round 4: lane 0 walked the 256 symbols eight times over -- widths, Kraft sum, weight statistics, the
// decoder table, the state chains' bit writer, the raw form, the canonical code values -- 156 us a launch however few segments, most of
// it dependent LDS round trips; tools/ktrace: cfg4 at 1 250 chunks.)  Lane t owns symbols 4 t .. 4 t + 3; reductions are wave
// reductions, histograms and ranks are ballots, the bit writer of the state chains is a prefix sum of the widths + LDS ORs.  What stays
// serial: the NCount header (<= 13 symbols) and the two state chains (each state is found from the one two weights later).
__device__ __forceinline__ uint32_t wave_sum32(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d);
    return v;
}
__device__ __forceinline__ uint32_t wave_max32(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)v, d); v = o > v ? o : v; }
    return v;
}

__global__ void __launch_bounds__(64) huf0_table_kernel(const uint8_t* __restrict__ tables, uint8_t* __restrict__ recs)
{
    __shared__ uint8_t lens[256], w[256], hdr[192], tsym[64], tnb[64], D[256];
    __shared__ __attribute__((aligned(4))) uint8_t f[320];
    __shared__ uint16_t tnew[64], vals[256];
    __shared__ uint32_t s_hl;
    const int t = threadIdx.x;
    const uint64_t seg = blockIdx.x;
    const uint64_t lt_mask = t ? (~0ull >> (64 - t)) : 0ull;        // lanes below this one
    uint32_t ln[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const int s = 4 * t + e;
        ln[e] = (uint32_t)((tables[seg * 128 + (s >> 1)] >> (4 * (s & 1))) & 15u);
        lens[s] = (uint8_t)ln[e];
    }
    for (int k = t; k < 80; k += 64) ((uint32_t*)f)[k] = 0;
    if (t == 0) s_hl = 0;
    // ---- widths -> weights (HUF_compress: weight = tableLog + 1 - length), the checks of oracle_huf0_compress_batch
    uint32_t nz_l = 0, tl_l = 0, ms_l = 0;
#pragma unroll
    for (int e = 0; e < 4; e++) if (ln[e]) { nz_l++; tl_l = ln[e] > tl_l ? ln[e] : tl_l; ms_l = (uint32_t)(4 * t + e) + 1u; }
    const uint32_t nz = wave_sum32(nz_l), tl = wave_max32(tl_l), max_sym1 = wave_max32(ms_l);      // max_sym + 1 (0: none)
    uint32_t kr_l = 0;
#pragma unroll
    for (int e = 0; e < 4; e++) if (ln[e]) kr_l += 1u << (tl - ln[e]);
    const uint32_t kraft = wave_sum32(kr_l);
    const bool ok = nz >= 2 && tl <= 11 && kraft == (1u << tl);
    const uint32_t nw = ok ? max_sym1 - 1u : 0u;
    uint32_t wt[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const uint32_t s = (uint32_t)(4 * t + e);
        wt[e] = (ok && s < nw && ln[e]) ? tl + 1u - ln[e] : 0u;
        w[s] = (uint8_t)wt[e];
    }
    __syncthreads();
    // ---- FSE statistics of the weights, NCount, decoder table (fse_write_weights)
    uint32_t count[16];
    int norm[16];
    uint32_t maxw = 0;
    bool fse = ok && nw >= 2;
#pragma unroll
    for (int k = 0; k < 16; k++) { count[k] = 0; norm[k] = 0; }
    if (fse) {                                                         // (wave-uniform)
#pragma unroll
        for (int k = 0; k < 13; k++) {
            uint32_t c = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) c += (uint32_t)__popcll(__ballot((uint32_t)(4 * t + e) < nw && wt[e] == (uint32_t)k));
            count[k] = c;
            if (c) maxw = (uint32_t)k;
        }
        uint32_t present = 0, top = 0;
        for (uint32_t s = 0; s <= maxw; s++) { present += count[s] != 0; if (count[s] > count[top]) top = s; }
        int sum = 0;
        for (uint32_t s = 0; s <= maxw; s++) if (count[s]) { norm[s] = (int)((uint64_t)count[s] * 64u / nw); if (norm[s] < 1) norm[s] = 1; sum += norm[s]; }
        norm[top] += 64 - sum;
        if (present < 2 || norm[top] < 1) fse = false;
    }
    uint32_t hl = 0;
    if (fse) {
        // the NCount header: <= 13 symbols, every lane the same arithmetic, lane 0 writes
        uint64_t acc = 0;
        int nbits = 0;
        uint32_t nout = 0;
        auto add = [&](uint32_t v, int nb) {
            acc |= (uint64_t)v << nbits;
            nbits += nb;
            while (nbits >= 8) { if (t == 0) f[nout] = (uint8_t)acc; nout++; acc >>= 8; nbits -= 8; }
        };
        add(6 - 5, 4);
        int remaining = 65, threshold = 64, nb = 7;
        bool previous0 = false;
        uint32_t sym = 0;
        while (sym <= maxw && remaining > 1) {
            if (previous0) {
                uint32_t start = sym;
                while (sym <= maxw && !norm[sym]) sym++;
                if (sym > maxw) { fse = false; break; }
                while (sym >= start + 24) { start += 24; add(0xffff, 16); }
                while (sym >= start + 3) { start += 3; add(3, 2); }
                add(sym - start, 2);
            }
            int c = norm[sym++];
            const int max = (2 * threshold - 1) - remaining;
            remaining -= c;
            c++;
            if (c >= threshold) c += max;
            add((uint32_t)c, nb - (c < max));
            previous0 = c == 1;
            if (remaining < 1) { fse = false; break; }
            while (remaining < threshold) { nb--; threshold >>= 1; }
        }
        if (remaining != 1) fse = false;
        if (fse) {
            if (nbits > 0) { if (t == 0) f[nout] = (uint8_t)acc; nout++; }
            hl = nout;
            // the decoder table: step i of the symbol spread (cells (43 i) & 63 in symbol order) belongs to lane i
            uint32_t c2 = 0, my = 0;
            for (uint32_t s2 = 0; s2 <= maxw; s2++) { if ((uint32_t)t >= c2 && (uint32_t)t < c2 + (uint32_t)norm[s2]) my = s2; c2 += (uint32_t)norm[s2]; }
            tsym[(43u * (uint32_t)t) & 63u] = (uint8_t)my;
        }
    }
    __syncthreads();
    if (fse) {
        // cell t: its symbol's next state number = norm + (cells of the same symbol below it)
        const uint32_t my = tsym[t];
        uint32_t ns = 0;
        for (uint32_t s2 = 0; s2 <= maxw; s2++) {
            const uint64_t m = __ballot(my == s2);
            if (my == s2) ns = (uint32_t)norm[s2] + (uint32_t)__popcll(m & lt_mask);
        }
        const uint32_t nbv = 6u - (31u - (uint32_t)__clz((int)ns));
        tnb[t] = (uint8_t)nbv;
        tnew[t] = (uint16_t)((ns << nbv) - 64u);
    }
    if (t == 0) s_hl = fse ? hl : 0u;
    __syncthreads();
    // ---- the two state chains, last weight first: D[i] = decoder state when weight i is emitted; every lane
    // tests one table entry, the lowest match wins (= the symbol's smallest state at the chain ends)
    hl = s_hl;
    bool chain_ok = hl != 0;
    if (hl != 0) {
        const uint32_t my_sym = tsym[t], lo = tnew[t], hi = (uint32_t)tnew[t] + (1u << tnb[t]);
        uint32_t later[2] = {0, 0};                      // D[i + 2] of either parity
        uint32_t wi = nw ? w[nw - 1] : 0u;               // (the next weight is requested before this one's ballot)
        for (int i = (int)nw - 1; i >= 0; i--) {
            const uint32_t wnext = i > 0 ? w[i - 1] : 0u;
            const uint32_t target = later[i & 1];
            const bool match = my_sym == wi && (i + 2 >= (int)nw || (target >= lo && target < hi));
            const uint64_t m = __ballot(match);
            if (m == 0) { chain_ok = false; break; }
            const uint32_t found = (uint32_t)__builtin_ctzll(m);
            later[i & 1] = found;
            if (t == 0) D[i] = (uint8_t)found;
            wi = wnext;
        }
    }
    __syncthreads();
    uint32_t fs = 0;
    if (ok && hl != 0 && chain_ok) {
        // the chains' bits behind the NCount bytes: items j = 0 .. nw - 3 are weights i = nw - 3 - j (value D[i + 2] - tnew[D[i]], tnb[D[i]]
        // bits), then D[1] and D[0] (6 bits each), then the end mark; positions = a prefix sum, bits OR-ed into the zeroed buffer
        const uint32_t nitems = nw >= 2 ? nw - 2u : 0u;
        uint32_t v[4], nbv[4], mine = 0;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const uint32_t j = (uint32_t)(4 * t + e);
            v[e] = 0; nbv[e] = 0;
            if (j < nitems) {
                const uint32_t i = nitems - 1u - j, d = D[i];
                v[e] = (uint32_t)D[i + 2] - (uint32_t)tnew[d];
                nbv[e] = tnb[d];
            }
            mine += nbv[e];
        }
        uint32_t inc = mine;
#pragma unroll
        for (int d2 = 1; d2 < 64; d2 <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)inc, d2); if (t >= d2) inc += u; }
        const uint32_t total = (uint32_t)__shfl((int)inc, 63);
        uint32_t at = 8u * hl + inc - mine;
        auto or_bits = [&](uint32_t bp, uint32_t val, uint32_t nb2) {
            if (nb2 == 0) return;
            const uint64_t x = (uint64_t)val << (bp & 31u);
            uint32_t* const q = (uint32_t*)f + (bp >> 5);
            atomicOr(q, (uint32_t)x);
            if ((uint32_t)(x >> 32)) atomicOr(q + 1, (uint32_t)(x >> 32));
        };
#pragma unroll
        for (int e = 0; e < 4; e++) { or_bits(at, v[e] & ((1u << nbv[e]) - 1u), nbv[e]); at += nbv[e]; }
        if (t == 0) {
            const uint32_t p = 8u * hl + total;
            or_bits(p, D[1], 6);
            or_bits(p + 6, D[0], 6);
            or_bits(p + 12, 1, 1);                        // BIT_closeCStream's end mark
        }
        fs = hl + (total + 13u + 7u) / 8u;
    }
    __syncthreads();
    const uint32_t raw = nw <= 128 ? 1 + (nw + 1) / 2 : 0;
    uint32_t hlen = 0;
    if (ok) {
        if (fs > 1 && fs < 128 && (raw == 0 || fs + 1 < raw)) {
            if (t == 0) hdr[0] = (uint8_t)fs;
            for (uint32_t k = (uint32_t)t; k < fs; k += 64) hdr[1 + k] = f[k];
            hlen = fs + 1;
        } else if (raw != 0 && nw != 0) {
            if (t == 0) hdr[0] = (uint8_t)(127 + nw);
            for (uint32_t k = 2u * (uint32_t)t; k < nw; k += 128) hdr[1 + k / 2] = (uint8_t)((w[k] << 4) | (k + 1 < nw ? w[k + 1] : 0));
            hlen = raw;
        }
        // Huff0's canonical code values (HUF_buildCTable): per length ascending symbols, the longest codes lowest
        uint32_t start[16];
        {
            uint32_t per[16];
#pragma unroll
            for (int l = 0; l < 16; l++) {
                uint32_t c = 0;
#pragma unroll
                for (int e = 0; e < 4; e++) c += (uint32_t)__popcll(__ballot(ln[e] == (uint32_t)l));
                per[l] = c;
                start[l] = 0;
            }
            uint32_t min = 0;
            for (uint32_t l = tl; l > 0; l--) { start[l] = min; min += per[l]; min >>= 1; }
        }
#pragma unroll
        for (int e = 0; e < 4; e++) vals[4 * t + e] = 0;
        for (uint32_t l = 1; l <= tl; l++) {
            uint64_t m[4];
#pragma unroll
            for (int e = 0; e < 4; e++) m[e] = __ballot(ln[e] == l);
            // symbols of this length below 4 t + e: all four of every lower lane's, and this lane's own lower ones
            uint32_t below = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) below += (uint32_t)__popcll(m[e] & lt_mask);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                if (ln[e] == l) vals[4 * t + e] = (uint16_t)(start[l] + below);
                below += ln[e] == l ? 1u : 0u;
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; e++) vals[4 * t + e] = 0;
    }
    __syncthreads();
    uint8_t* const rec = recs + seg * kRecBytes;
    if (t == 0) { ((uint32_t*)rec)[0] = hlen; ((uint32_t*)rec)[1] = ok ? tl : 0u; }
    for (int k = t; k < 160; k += 64) rec[kRecHdr + k] = k < (int)hlen ? hdr[k] : (uint8_t)0;
    for (int k = t; k < 256; k += 64) {
        ((uint16_t*)(rec + kRecTab))[k] = hlen ? vals[k] : (uint16_t)0;
        rec[kRecTab + 512 + k] = lens[k];
    }
}

// H2.  One workgroup per segment = 64 chunks x 4 streams; lane = (chunk, stream).
// meta[c] = bytes0 | bytes1 << 16 | bytes2 << 32 | mode << 48   (mode 0 empty, 1 stored, 2 repeated byte, 3 coded)
__global__ void __launch_bounds__(256) huf0_size_kernel(const uint8_t* __restrict__ dense, const uint64_t* __restrict__ offsets,
                                                        const uint32_t* __restrict__ sizes, uint64_t nchunks,
                                                        const uint8_t* __restrict__ recs, uint32_t* __restrict__ bsizes,
                                                        uint64_t* __restrict__ meta)
{
    __shared__ uint8_t lens[256];
    // The pass reads every stream once, 64 bytes a lane per trip, and needs no LDS to speak of -- so 32 waves a CU were resident, 65 000 lanes per
    // XCD each with its own 128-byte line open, and a line had left the 4 MB L2 again before its lane came back for the second half
    // (FETCH_SIZE x 2 = 4.4 GB for 2.9 GB of streams).  For big batches the launch claims kSizePassPad bytes of dynamic LDS the kernel does
    // not use: 16 waves a CU, the writer's four passes 6.08 -> 5.65 ms at 800 000 chunks in the first same-box run (8 waves: the same; 24: no
    // gain), 6.0 -> 5.7 .. 6.0 in later ones.  Not below a few rounds of workgroups: at 80 000 chunks (1 250 workgroups) half the resident slots
    // cost more than the L2 gives back (0.85 -> 0.91 ms).
    const int t = threadIdx.x, j = t & 3;
    const uint64_t seg = blockIdx.x, c = seg * SEG + (uint64_t)(t >> 2);
    const uint8_t* const rec = recs + seg * kRecBytes;
    const uint32_t hlen = ((const uint32_t*)rec)[0];
    lens[t] = rec[kRecTab + 512 + t];
    __syncthreads();
    const bool exists = c < nchunks;
    const uint32_t n = exists ? sizes[c] : 0u;
    const uint8_t* const s = dense + (exists ? offsets[c] : 0);
    uint32_t k0, k1;
    sub_range(n, j, k0, k1);
    if (j == 3) k1 = n;
    const uint32_t first = n ? s[0] : 0u;
    uint32_t bits = 0;
    bool same = true;
    uint32_t k = k0;
    auto sixteen = [&](const u32x4& x) {
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const uint32_t v = d == 0 ? x.x : d == 1 ? x.y : d == 2 ? x.z : x.w;
            bits += lens[v & 255] + lens[(v >> 8) & 255] + lens[(v >> 16) & 255] + lens[v >> 24];
            same = same && v == first * 0x01010101u;
        }
    };
    // four pieces in flight: a lane's loads are otherwise one dependent round trip to memory per 16 bytes
    for (; k + 64 <= k1; k += 64) {
        const u32x4 x0 = *(const u32x4_a1*)(s + k), x1 = *(const u32x4_a1*)(s + k + 16), x2 = *(const u32x4_a1*)(s + k + 32),
                    x3 = *(const u32x4_a1*)(s + k + 48);
        sixteen(x0); sixteen(x1); sixteen(x2); sixteen(x3);
    }
    for (; k + 16 <= k1; k += 16) { const u32x4 x = *(const u32x4_a1*)(s + k); sixteen(x); }
    for (; k < k1; k++) { bits += lens[s[k]]; same = same && s[k] == first; }
    const uint32_t bytes = (bits + 1 + 7) >> 3;
    const int all_same = __builtin_amdgcn_mov_dpp((int)same, 0x00, 0xf, 0xf, true) & __builtin_amdgcn_mov_dpp((int)same, 0x55, 0xf, 0xf, true) &
                         __builtin_amdgcn_mov_dpp((int)same, 0xAA, 0xf, 0xf, true) & __builtin_amdgcn_mov_dpp((int)same, 0xFF, 0xf, 0xf, true);
    const uint32_t b0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)bytes, 0x00, 0xf, 0xf, true), b1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)bytes, 0x55, 0xf, 0xf, true),
                   b2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)bytes, 0xAA, 0xf, 0xf, true), b3 = (uint32_t)__builtin_amdgcn_mov_dpp((int)bytes, 0xFF, 0xf, 0xf, true);
    if (exists && j == 0) {
        uint32_t mode, size;
        const uint64_t total = (uint64_t)hlen + 6 + b0 + b1 + b2 + b3;
        if (n == 0) { mode = 0; size = 0; }
        else if (all_same) { mode = 2; size = 1; }
        else if (hlen != 0 && n >= 12 && b0 <= 65535 && b1 <= 65535 && b2 <= 65535 && total < n) { mode = 3; size = (uint32_t)total; }
        else { mode = 1; size = n; }
        bsizes[c] = size;
        meta[c] = (uint64_t)(b0 & 0xffffu) | ((uint64_t)(b1 & 0xffffu) << 16) | ((uint64_t)(b2 & 0xffffu) << 32) | ((uint64_t)mode << 48);
    }
}

// H3.  Same mapping.  A stream is written last symbol first; its bytes collect in a per-lane LDS ring
// ring[dword][lane] and leave as 64-byte units (four 16-byte stores back to back), the rest narrow.
__global__ void __launch_bounds__(256) huf0_encode_kernel(const uint8_t* __restrict__ dense, const uint64_t* __restrict__ offsets,
                                                          const uint32_t* __restrict__ sizes, uint64_t nchunks,
                                                          const uint8_t* __restrict__ recs, const uint64_t* __restrict__ meta,
                                                          uint8_t* __restrict__ out, const uint64_t* __restrict__ boffs)
{
#ifndef HUF0W_RING
#define HUF0W_RING 64                     // dwords of output ring a lane; a flush is half of it (128 bytes: 64 KB a workgroup, 8 waves a CU)
#endif
    constexpr uint32_t kRD = HUF0W_RING, kRM = kRD - 1u, kFlush = kRD / 2u;
    __shared__ uint32_t ring[kRD * 256];
    __shared__ uint32_t tab[256];                        // val | len << 16
    __shared__ uint8_t hdr[160];
#ifndef HUF0W_PAD
#define HUF0W_PAD 0                       // experiment: extra LDS bytes a workgroup claims (fewer resident waves, fewer open lines per L2)
#endif
#if HUF0W_PAD
    __shared__ volatile uint8_t pad_[HUF0W_PAD];
    pad_[threadIdx.x & 3] = 0;                           // (volatile: keeps the array)
#endif
    const int t = threadIdx.x, j = t & 3;
    const uint64_t seg = blockIdx.x, c = seg * SEG + (uint64_t)(t >> 2);
    const uint8_t* const rec = recs + seg * kRecBytes;
    const uint32_t hlen = ((const uint32_t*)rec)[0];
    tab[t] = (uint32_t)((const uint16_t*)(rec + kRecTab))[t] | ((uint32_t)rec[kRecTab + 512 + t] << 16);
    if (t < 160) hdr[t] = rec[kRecHdr + t];
    __syncthreads();
    if (c >= nchunks) return;
    const uint32_t n = sizes[c];
    const uint8_t* const s = dense + offsets[c];
    uint8_t* const o = out + boffs[c];
    const uint64_t m = meta[c];
    const uint32_t mode = (uint32_t)(m >> 48);
    if (mode == 0) return;
    if (mode == 2) { if (j == 0) o[0] = s[0]; return; }
    if (mode == 1) {                                      // stored: 16-byte pieces over the quad, then the odd bytes
        uint32_t k = (uint32_t)j * 16u;
        for (; k + 16 <= n; k += 64) *(u32x4_a1*)(o + k) = *(const u32x4_a1*)(s + k);
        for (uint32_t r = (n & ~15u) + (uint32_t)j; r < n; r += 4) o[r] = s[r];
        return;
    }
    const uint32_t b0 = (uint32_t)(m & 0xffffu), b1 = (uint32_t)((m >> 16) & 0xffffu), b2 = (uint32_t)((m >> 32) & 0xffffu);
    for (uint32_t k = (uint32_t)j; k < hlen; k += 4) o[k] = hdr[k];
    if (j == 0) {
        uint8_t* const jt = o + hlen;
        jt[0] = (uint8_t)b0; jt[1] = (uint8_t)(b0 >> 8); jt[2] = (uint8_t)b1; jt[3] = (uint8_t)(b1 >> 8); jt[4] = (uint8_t)b2; jt[5] = (uint8_t)(b2 >> 8);
    }
    uint8_t* const so = o + hlen + 6 + (j > 0 ? b0 : 0u) + (j > 1 ? b1 : 0u) + (j > 2 ? b2 : 0u);
    uint32_t k0, k1;
    sub_range(n, j, k0, k1);
    if (j == 3) k1 = n;
    uint32_t* const my = ring + t;
    uint64_t acc = 0;
    uint32_t nbits = 0, wd = 0, fd = 0;                   // dwords appended to / flushed from the ring
    auto put = [&](uint32_t sym) {
        const uint32_t e = tab[sym];
        acc |= (uint64_t)(e & 0xffffu) << nbits;
        nbits += e >> 16;
    };
    auto drain = [&]() {                                  // whole dwords of the accumulator -> ring; whole 64-byte units -> HBM
        if (nbits >= 32) {
            my[(wd & kRM) << 8] = (uint32_t)acc;
            wd++;
            acc >>= 32;
            nbits -= 32;
            if (wd - fd >= kFlush) {
                const uint32_t d0 = fd & kRM;
#pragma unroll
                for (int q = 0; q < (int)kFlush / 4; q++) {
                    const u32x4 p = {my[(d0 + 4 * q) << 8], my[(d0 + 4 * q + 1) << 8], my[(d0 + 4 * q + 2) << 8], my[(d0 + 4 * q + 3) << 8]};
                    *(u32x4_a1*)(so + 4u * fd + 16u * q) = p;
                }
                fd += kFlush;
            }
        }
    };
    // the ragged end first, byte by byte, then 16 source bytes per load (last symbol first)
    uint32_t k = k1;
    const uint32_t r = (k1 - k0) & 15u;
    for (uint32_t i = 0; i < r; i++) { put(s[--k]); drain(); }
    // 16 symbols per trip.  Their 16 table reads do not depend on the bit accumulator and go out together; the accumulator chain
    // is pure VALU; a dword leaves for the ring after every second symbol WITHOUT a branch (the slot is written every time and
    // only counted when 32 bits are there: a slot written early is written again when it is due); the 64-byte flush to memory
    // is looked at once per trip (<= 6 dwords a trip join the < half a ring waiting).  The loop used to carry eight
    // conditional drains a trip, each with its own flush -- 45 s_waitcnt and 15 branches per 16 symbols, 4.5 ms at 800 000
    // chunks for 13 VALU a symbol.
    auto trip16 = [&](const u32x4& x) {                   // 16 symbols, last first (the comment above)
        uint32_t e[16];
#pragma unroll
        for (int d = 3; d >= 0; d--) {
            const uint32_t v = d == 0 ? x.x : d == 1 ? x.y : d == 2 ? x.z : x.w;
            e[4 * (3 - d) + 0] = tab[v >> 24];
            e[4 * (3 - d) + 1] = tab[(v >> 16) & 255u];
            e[4 * (3 - d) + 2] = tab[(v >> 8) & 255u];
            e[4 * (3 - d) + 3] = tab[v & 255u];
        }
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
            acc |= (uint64_t)(e[j] & 0xffffu) << nbits;
            nbits += e[j] >> 16;
            acc |= (uint64_t)(e[j + 1] & 0xffffu) << nbits;
            nbits += e[j + 1] >> 16;
            my[(wd & kRM) << 8] = (uint32_t)acc;
            const bool full = nbits >= 32;
            wd += full ? 1u : 0u;
            acc = full ? acc >> 32 : acc;
            nbits -= full ? 32u : 0u;
        }
        if (wd - fd >= kFlush) {
            const uint32_t d0 = fd & kRM;
#pragma unroll
            for (int q = 0; q < (int)kFlush / 4; q++) {
                const u32x4 pq = {my[(d0 + 4 * q) << 8], my[(d0 + 4 * q + 1) << 8], my[(d0 + 4 * q + 2) << 8], my[(d0 + 4 * q + 3) << 8]};
                *(u32x4_a1*)(so + 4u * fd + 16u * q) = pq;
            }
            fd += kFlush;
        }
    };
#ifndef HUF0W_TRIP64
#define HUF0W_TRIP64 1
#endif
#if HUF0W_TRIP64
    // The source is read 64 bytes a lane at a time, one trip ahead: with 16-byte loads a lane came back to the same 128-byte line eight
    // times, ~45 000 lanes per XCD between two visits, and the line had left the L2 again (FETCH_SIZE x 2 = 13.6 GB for 2.9 GB of streams).
    while (((k - k0) & 63u) != 0u) {                      // down to a whole number of 64-byte trips
        k -= 16;
        const u32x4 x = *(const u32x4_a1*)(s + k);
        trip16(x);
    }
    if (k > k0) {
        u32x4 nx[4];
#pragma unroll
        for (int q = 0; q < 4; q++) nx[q] = *(const u32x4_a1*)(s + k - 16u * (q + 1));
        while (k > k0) {
            k -= 64;
            u32x4 x4[4];
#pragma unroll
            for (int q = 0; q < 4; q++) x4[q] = nx[q];
            if (k > k0) {                                     // the next trip's 64 bytes, requested before this trip's are coded
#pragma unroll
                for (int q = 0; q < 4; q++) nx[q] = *(const u32x4_a1*)(s + k - 16u * (q + 1));
            }
#pragma unroll
            for (int q = 0; q < 4; q++) trip16(x4[q]);
        }
    }
#else
    while (k > k0) {
        k -= 16;
        const u32x4 x = *(const u32x4_a1*)(s + k);
        trip16(x);
    }
#endif
    acc |= 1ull << nbits;                                 // the closing 1 bit (BIT_closeCStream)
    nbits += 1;
    drain();
    // what is left: ring dwords [fd, wd), then the accumulator's bytes
    for (; fd < wd; fd++) *(u32_any*)(so + 4u * fd) = my[(fd & kRM) << 8];
    uint8_t* tail = so + 4u * wd;
    for (uint32_t b = 0; b < nbits; b += 8) *tail++ = (uint8_t)(acc >> b);
}

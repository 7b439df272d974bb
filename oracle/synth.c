/*
 * synth.c -- the deterministic synthetic inputs of SURVEY.md 8(d), C side.  TEST/BENCH
 * INFRASTRUCTURE ONLY (lives in liboracle.so).  The torch twin is tools/synth.py; tests/
 * test_synth_cpu.py holds them to the same bytes.
 *
 * Every chunk c is an independent series drawn from splitmix64 seeded with seed ^ c:
 *     draw(c, i) = mix64((seed ^ c) + (i + 1) * 0x9E3779B97F4A7C15)        i = 0, 1, 2, ...
 *     u32(c, i)  = draw(c, i) >> 32
 * For rows x ndims samples of w bits:
 *   G0 "uniform":  x[r][d] = u32(c, r*D + d) >> (32 - w)
 *   G1 "walk" s :  x[0][d] = u32(c, d) >> (32 - w);
 *                  x[r][d] = x[r-1][d] + ((u32(c, r*D + d) * (2s+1)) >> 32) - s     (mod 2^w), r >= 1
 *   G2 "walkflat": G1, but the step is 0 in rows with (r / 64) % 4 == 0 (runs for the RLE stage)
 */
#include <stdint.h>
#include <stddef.h>

static uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* kind: 0 uniform, 1 walk, 2 walkflat; out: nchunks * rows * ndims elements of elem_bytes */
void synth_fill(int kind, int elem_bytes, uint64_t seed, uint64_t chunk0, uint64_t nchunks, uint32_t rows, uint32_t ndims,
                uint32_t step, void* out)
{
    const int w = 8 * elem_bytes;
    const uint32_t mask = (uint32_t)((1ull << w) - 1);
    uint8_t* o8 = (uint8_t*)out;
    uint16_t* o16 = (uint16_t*)out;
    uint32_t prev[4096];
    if (ndims > 4096) return;
    for (uint64_t c = 0; c < nchunks; c++) {
        const uint64_t s0 = seed ^ (chunk0 + c);
        const size_t base = (size_t)c * rows * ndims;
        for (uint32_t r = 0; r < rows; r++) {
            for (uint32_t d = 0; d < ndims; d++) {
                const uint64_t i = (uint64_t)r * ndims + d;
                const uint32_t u = (uint32_t)(mix64(s0 + (i + 1) * 0x9E3779B97F4A7C15ull) >> 32);
                uint32_t x;
                if (kind == 0 || r == 0) {
                    x = u >> (32 - w);
                } else {
                    int64_t st = (int64_t)(((uint64_t)u * (2ull * step + 1)) >> 32) - (int64_t)step;
                    if (kind == 2 && (r / 64) % 4 == 0) st = 0;
                    x = (uint32_t)((int64_t)prev[d] + st) & mask;
                }
                prev[d] = x;
                if (elem_bytes == 1) o8[base + i] = (uint8_t)x;
                else o16[base + i] = (uint16_t)x;
            }
        }
    }
}

/*
 * The paper's full chain on host cores, chunk by chunk, for bench.py's cpu_baseline of BASELINE
 * config 4: an entropy decoder (Huff0: the system libzstd's HUF_decompress, or oracle_huf0_decompress)
 * followed by a Sprintz decoder (the compiled reference's ref_decompress, or oracle_decompress),
 * both handed in as function pointers by the caller so that this file links against neither.
 */
typedef size_t (*huf_fn)(void* dst, size_t dst_size, const void* src, size_t src_size);
typedef int64_t (*dec_fn)(int codec, int elem_bytes, const void* src, void* dest);
uint64_t oracle_huf0_chain_chunks(void* huf, void* dec, int codec, int elem_bytes, const uint8_t* blocks, const uint64_t* block_offsets,
                                  const uint32_t* stream_sizes, uint64_t nchunks, uint32_t chunk_len, uint8_t* scratch, void* out)
{
    uint8_t* o = (uint8_t*)out;
    uint64_t total = 0;
    for (uint64_t c = 0; c < nchunks; c++) {
        ((huf_fn)huf)(scratch, stream_sizes[c], blocks + block_offsets[c], (size_t)(block_offsets[c + 1] - block_offsets[c]));
        const int64_t n = ((dec_fn)dec)(codec, elem_bytes, scratch, o + c * (uint64_t)chunk_len * (uint64_t)elem_bytes);
        if (n > 0) total += (uint64_t)n;
    }
    return total;
}

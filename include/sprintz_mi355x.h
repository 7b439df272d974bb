/*
 * sprintz_mi355x.h -- C-ABI of libsprintz_mi355x.so: the Sprintz codec hot path
 * (forecast -> zigzag -> per-block nbits -> bit-pack -> RLE) on AMD MI355X
 * (gfx950), bit-exact with dblalock/sprintz cpp/Compress.
 *
 * Plain C: pointers and sizes only.  Two groups of entry points:
 *
 *  (1) Drop-in single-call API.  One symbol per reference function of
 *      cpp/Compress/sprintz.h:16-32, same argument meaning, same return
 *      values (ELEMENTS, -1 for ndims == 0, floor'ed for odd 16-bit byte
 *      lengths -- sprintz_xff_rle.cpp:554), HOST pointers in and out.
 *      The reference's functions have C++ linkage (default argument
 *      `write_size=true`): the library ALSO exports them under exactly those
 *      C++ signatures, i.e. the reference's own mangled names
 *      (sprintz_amd/csrc/dropin.cpp, declared in include/sprintz_dropin.hpp),
 *      so an object compiled against the reference's sprintz.h -- e.g. the
 *      lzbench fork, reference README.md:29 -- links against this library
 *      unchanged.  One call = one chunk = one wavefront's worth of work: it
 *      is correct, not fast.  See (2).
 *
 *  (2) Batched device API.  The unit of GPU parallelism is the CHUNK: an
 *      independent compress() call on a contiguous slice, exactly what
 *      lzbench's block-size option does (reference README.md:58 "1KB and
 *      10KB blocks").  Predictor state resets per chunk, so chunks are
 *      embarrassingly parallel.  DEVICE pointers, explicit hipStream_t
 *      (passed as void* so that this header needs no HIP include).
 *
 * All functions return 0 / a non-negative count on success and a negative
 * SPRINTZ_E_* code on failure.  There is NO CPU fallback anywhere in this
 * library: without a usable HIP device every entry point fails with
 * SPRINTZ_E_NO_DEVICE.
 */
#ifndef SPRINTZ_MI355X_H
#define SPRINTZ_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history (additive: a caller built against version n runs against any library with abi_version() >= n)
 *   1  round 1: the 8 drop-in entry points, the batched device API, the optional Huffman ("SPZH") and query stages
 *   2  round 2: set_option, *_layout, the Huff0 wire format (huf0_*), online_*, comm_* / gather_layout / layout_bases,
 *      the column-major and run-less codec entry points, the reference's mangled C++ names (sprintz_dropin.hpp)
 *   3  round 3: compress_batch_dense / compress_dense_tmp_bytes, SPRINTZ_OPT_DENSE_MODE, env SPRINTZ_MI355X_RCCL_SONAME
 *   4  round 3: compress_batch_colmajor_dense, SPRINTZ_OPT_SPLIT_LANES, SPRINTZ_OPT_ENC_PAIR
 *   5  round 4: SPRINTZ_OPT_HOST_WAIT, SPRINTZ_OPT_LAT_CHUNKS, SPRINTZ_OPT_HOST_STREAMS, SPRINTZ_OPT_REF_DECODER_QUIRK (the single-call entry points work on a mapped staging buffer: one wait per call)
 *   6  round 5: huf0_decompress_batch_hint, SPRINTZ_OPT_HUF0_SYNC_CHUNKS, SPRINTZ_MI355X_MAX_NDIMS 65535
 *   7  round 6: SPRINTZ_OPT_BLK_CHUNKS (block-parallel delta kernels); the batched entry points refuse shapes whose tail outgrows remaining_len */
#define SPRINTZ_MI355X_ABI_VERSION 7

/* codec ids */
#define SPRINTZ_CODEC_DELTA 0   /* sprintz_*_delta_*  (sprintz_delta_rle.cpp / sprintz_delta_lowdim.cpp) */
#define SPRINTZ_CODEC_XFF   1   /* sprintz_*_xff_*    (sprintz_xff_rle.cpp   / sprintz_xff_lowdim.cpp), FIRE */
/* the reference's older codecs without run-length coding: 6-byte header
 * {u32 len; u16 ndims}, general row-major layout for every ndims; accepted by every batched
 * entry point; decoded by the generic kernels */
#define SPRINTZ_CODEC_DELTA_NORLE   2   /* compress_rowmajor_delta_{8b,16b}  sprintz_delta.cpp:776-1391 */
#define SPRINTZ_CODEC_BITPACK_NORLE 3   /* compress_rowmajor_{8b,16b}: bit-packing only  sprintz_delta.cpp:64-773 */
#define SPRINTZ_CODEC_XFF_NORLE     4   /* compress8b_rowmajor_xff (8-bit only; 8-byte header)  sprintz_xff.cpp:35-626 */

/* error codes */
#define SPRINTZ_E_INVALID    (-1)   /* bad argument; also what the reference returns for ndims == 0 (sprintz.cpp:36) */
#define SPRINTZ_E_NO_DEVICE  (-2)   /* no HIP device / HIP runtime error at init */
#define SPRINTZ_E_HIP        (-3)   /* a HIP call failed (see sprintz_mi355x_last_error) */
#define SPRINTZ_E_UNSUPPORTED (-4)  /* ndims above SPRINTZ_MI355X_MAX_NDIMS (or above 512 where only the lane-group kernels exist) */
#define SPRINTZ_E_CORRUPT    (-5)   /* decoder: stream header disagrees with the arguments */

#define SPRINTZ_MI355X_MAX_NDIMS 65535  /* whatever the stream header's uint16 holds (format.h:36-45).  513 .. 65535: the four RLE codec pairs, row-major,
                                           no query (csrc/any_ndims.hip: one workgroup per chunk; from 2048 on in column tiles); everything else <= 512 */

/* Extra readable bytes the decoder may touch after the last stream byte and
 * the encoder after the last input element (aligned 8-byte windows; the
 * reference has the same kind of contract). */
#define SPRINTZ_MI355X_READ_SLACK 16

int         sprintz_mi355x_abi_version(void);
const char* sprintz_mi355x_last_error(void);     /* thread-local, never NULL; describes the last call of
                                                    this thread that returned a negative code */

/* Tuning knobs (process-wide, atomic).  Their initial values come from the environment, read once:
 *   SPRINTZ_OPT_NO_FAST           1 = route every call to the generic kernels (A/B runs, tests);
 *                                 env SPRINTZ_MI355X_NO_FAST
 *   SPRINTZ_OPT_CHUNKS_PER_GROUP  consecutive chunks one lane group of the headline decoder walks,
 *                                 1..64, default 1; env SPRINTZ_MI355X_CHUNKS_PER_GROUP
 *   SPRINTZ_OPT_DENSE_MODE        how sprintz_mi355x_compress_batch_dense builds the container: 1 (default) = inside the
 *                                 encode launch (csrc/compact_tail.h), 0 = encode, then scan + copy (A/B runs, tests);
 *                                 env SPRINTZ_MI355X_DENSE_MODE
 *   SPRINTZ_OPT_HUF0_BIG_BATCH    chunks from which the Huff0 reader's one-table stream kernel runs as 2-wave workgroups with
 *                                 64-byte stream pieces (the built defaults HUF0_BIG_WG = 2, HUF0_BIG_PLOG = 6) instead of single waves,
 *                                 which are faster while each has a SIMD to itself (16 chunks a wave, 1 024 SIMDs); default 16385,
 *                                 0 = always (tests)
 *   SPRINTZ_OPT_HUF0_SYNC_CHUNKS  batches of at most this many chunks run the Huff0 reader's stream stage as one wave per chunk with sixteen
 *                                 self-synchronising decoders per stream (csrc/huf0_sync.h: a stream's ~900-look-up chain becomes three passes
 *                                 over ~60: what counts while the chip has idle issue slots -- 86 -> 45 us at 1 250 chunks, even at 10 000); default 8192, 0 = never (A/B runs, tests);
 *                                 env SPRINTZ_MI355X_HUF0_SYNC_CHUNKS
 *   SPRINTZ_OPT_SPLIT_LANES       1 (default) = 8-bit row-major streams of 65 .. 80 columns decode on 32 lanes a chunk (a pair of
 *                                 adjacent columns + one single column per lane, two chunks a wavefront), 0 = on 64 lanes x 2
 *                                 columns like the other shapes up to 128 columns; 0 also sizes the LDS carve of 16-bit streams of
 *                                 65 .. 80 columns for 128 columns again instead of 80 (A/B runs, tests); env SPRINTZ_MI355X_SPLIT_LANES
 *   SPRINTZ_OPT_ENC_PAIR          chunks from which row-major streams of 5 .. 64 columns are encoded with two columns per lane (the
 *                                 65 .. 128-column kernel on 4 .. 32 lanes a chunk: fewer instructions per sample) instead of one (a
 *                                 chunk's latency is shorter: what counts for a handful of chunks); default 1024, 1 = always,
 *                                 0 = never (A/B runs, tests); env SPRINTZ_MI355X_ENC_PAIR
 *   SPRINTZ_OPT_HOST_STREAMS      streams the single-call entry points of ALL host threads share per device (each call waits for its own
 *                                 launches through an event); default 4, 0 = a private stream per thread; read when a thread makes its
 *                                 first call; env SPRINTZ_MI355X_HOST_STREAMS
 *   SPRINTZ_OPT_REF_DECODER_QUIRK 0 (default) = every decoder is the true inverse of the reference ENCODER; 1 = runs of 16-bit
 *                                 general-layout FIRE streams are replayed as the reference DECODER replays them (coefficient
 *                                 shifted by 4 instead of 12, odd columns reading the counters' high halves:
 *                                 sprintz_xff_rle.cpp:893-901) -- not lossless on streams whose runs start with a non-zero
 *                                 prediction, but sample-for-sample what sprintz_decompress_xff_16b of the reference returns;
 *                                 env SPRINTZ_MI355X_REF_DECODER_QUIRK
 *   SPRINTZ_OPT_LAT_CHUNKS        batches of at most this many chunks (both layouts, 1 .. 64 columns, chunks of at most 16 KB; single calls and batches of
 *                                 at most 64 chunks up to ~40 KB of uint16 / ~24 KB of uint8: what fits a workgroup's 150 KB of LDS) decode
 *                                 with one workgroup per chunk (csrc/decode_lat.h: a chunk's latency is what counts; a third as many from 17 columns on); default 2048,
 *                                 0 = never (A/B runs, tests); env SPRINTZ_MI355X_LAT_CHUNKS
 *   SPRINTZ_OPT_BLK_CHUNKS        batches of at least this many chunks of the DELTA codec take the block-parallel kernels (a thread per block and
 *                                 16-byte row piece / per 16 rows of a univariate stream; csrc/encode_blk.h, decode_blk.h) where the shape allows;
 *                                 0 = never, default 2049 (env SPRINTZ_MI355X_BLK_CHUNKS).  Same bytes either way.
 *   SPRINTZ_OPT_BLK_KERNELS       which of the round-6 delta kernels such batches take, a mask: 1 = the block-parallel general-layout encoder, 2 = the
 *                                 block-parallel general-layout decoder, 4 = the block-parallel univariate low-dim encoder, 8 = the piece-sequential
 *                                 general-layout decoder (csrc/decode_row.h; wins over 2) on the shapes it measured faster on (8-bit rows of at
 *                                 least 32 columns), 16 = ... on every shape it fits (tests); default 9: the ones that measured faster than the
 *                                 lane-per-column kernels on BASELINE's configurations (env SPRINTZ_MI355X_BLK_KERNELS)
 *   SPRINTZ_OPT_HOST_WAIT         how a single-call entry point waits for its launches: 0 (default) = spin (hipStreamSynchronize)
 *                                 while at most 4 callers (and at most half of the CPUs this process may use) are inside the library,
 *                                 otherwise sleep and poll a mapped host word that a one-thread kernel at the end of the call writes
 *                                 (no runtime wait: 64 threads on 16 CPUs get 4x the calls per second of either runtime wait; a
 *                                 thread that waits this way has its timer slack at 1 us, prctl(PR_SET_TIMERSLACK), for the duration of the wait -- its own value is put back before the call returns); 1 = always
 *                                 spin; 2 = always sleep and poll; env SPRINTZ_MI355X_HOST_WAIT */
#define SPRINTZ_OPT_NO_FAST 0
#define SPRINTZ_OPT_CHUNKS_PER_GROUP 1
#define SPRINTZ_OPT_DENSE_MODE 2
#define SPRINTZ_OPT_HUF0_BIG_BATCH 3
#define SPRINTZ_OPT_SPLIT_LANES 4
#define SPRINTZ_OPT_ENC_PAIR 5
#define SPRINTZ_OPT_HOST_WAIT 6
#define SPRINTZ_OPT_LAT_CHUNKS 7
#define SPRINTZ_OPT_HOST_STREAMS 8
#define SPRINTZ_OPT_REF_DECODER_QUIRK 9
#define SPRINTZ_OPT_HUF0_SYNC_CHUNKS 10
#define SPRINTZ_OPT_BLK_CHUNKS 11
#define SPRINTZ_OPT_BLK_KERNELS 12
int sprintz_mi355x_set_option(int option, int value);

/* ------------------------------------------------------------------------
 * (1) Drop-in single-call API (host pointers).  Replaces, one to one:
 *   sprintz_compress_delta_8b    sprintz.h:18 / sprintz.cpp:57
 *   sprintz_decompress_delta_8b  sprintz.h:20 / sprintz.cpp:75
 *   sprintz_compress_xff_8b      sprintz.h:22 / sprintz.cpp:98
 *   sprintz_decompress_xff_8b    sprintz.h:24 / sprintz.cpp:115
 *   sprintz_compress_delta_16b   sprintz.h:28 / sprintz.cpp:138
 *   sprintz_decompress_delta_16b sprintz.h:30 / sprintz.cpp:156
 *   sprintz_compress_xff_16b     sprintz.h:32 / sprintz.cpp:180
 *   sprintz_decompress_xff_16b   sprintz.h:34 / sprintz.cpp:197
 * `dest` capacity: the reference's callers allocate len*3/2+64 elements for
 * compression and len+64 for decompression (test/compress_testing.hpp:145-147);
 * this implementation writes at most sprintz_mi355x_compress_bound() bytes /
 * exactly the decoded elements (it never over-runs like the reference does).
 * write_size == 0 omits the 8-byte header (sprintz_xff_rle.cpp:119-127).
 * Re-entrant from any number of host threads on disjoint buffers, like the reference (sprintz.h: one call = one thread).
 * Cost of a call (MI355X, 10 KB of uint16 x 8): 25 us either way -- memcpy into the calling thread's mapped staging buffer,
 * ONE launch (the one-workgroup-per-chunk kernel reads and writes that buffer directly and ends by writing the call's ticket
 * into a mapped host word), the caller polls that word (no runtime wait), memcpy out; chunks whose working set does not fit a
 * workgroup's LDS (above ~40 KB of uint16 / ~24 KB of uint8 to decode, ~34 / ~24 KB to encode) or of more than 64
 * columns take a staging kernel + the batched kernels' lane-per-column form + a wait (a chunk's latency then grows with its
 * groups: ~1.2 us each).  A decoder writes what the STREAM says (header counts, run lengths): like the reference's, these
 * entry points take no destination size, so a damaged stream can announce more samples than the caller's buffer holds --
 * callers that do not trust their input use the batched API, whose chunk_len bounds every chunk's output.
 * The batched API below is the fast path.
 * ---------------------------------------------------------------------- */
int64_t sprintz_mi355x_compress_delta_8b (const uint8_t*  src, uint32_t len, int8_t*  dest, uint16_t ndims, int write_size);
int64_t sprintz_mi355x_compress_xff_8b   (const uint8_t*  src, uint32_t len, int8_t*  dest, uint16_t ndims, int write_size);
int64_t sprintz_mi355x_compress_delta_16b(const uint16_t* src, uint32_t len, int16_t* dest, uint16_t ndims, int write_size);
int64_t sprintz_mi355x_compress_xff_16b  (const uint16_t* src, uint32_t len, int16_t* dest, uint16_t ndims, int write_size);

int64_t sprintz_mi355x_decompress_delta_8b (const int8_t*  src, uint8_t*  dest);
int64_t sprintz_mi355x_decompress_xff_8b   (const int8_t*  src, uint8_t*  dest);
int64_t sprintz_mi355x_decompress_delta_16b(const int16_t* src, uint16_t* dest);
int64_t sprintz_mi355x_decompress_xff_16b  (const int16_t* src, uint16_t* dest);

/* Headerless decode (the reference's 5-argument kernel form,
 * sprintz_xff.h:56-58 / sprintz_xff_rle.cpp:1181-1190): needed for streams
 * written with write_size == 0. */
int64_t sprintz_mi355x_decompress_noheader(int codec, int elem_bytes, const void* src, void* dest,
                                           uint16_t ndims, uint32_t ngroups, uint16_t remaining_len);

/* The layer below sprintz.h: the same four codecs with the payload layout chosen by the caller
 * instead of by ndims.  Replaces compress_rowmajor_{delta,xff}_rle[_lowdim]_{8b,16b} and their
 * 2-argument decoders (sprintz_delta.h:49-91, sprintz_xff.h:43-85):
 *   SPRINTZ_LAYOUT_AUTO     what sprintz.h's dispatch picks (sprintz.cpp:34-50)
 *   SPRINTZ_LAYOUT_GENERAL  row-major payload for every ndims (the *_rle_* names)
 *   SPRINTZ_LAYOUT_LOWDIM   column-major low-dim payload; ndims <= 4 @8b / <= 2 @16b, else -1
 *                           (sprintz_delta_lowdim.cpp:64-70)
 * codec: SPRINTZ_CODEC_DELTA or SPRINTZ_CODEC_XFF. */
#define SPRINTZ_LAYOUT_AUTO 0
#define SPRINTZ_LAYOUT_GENERAL 1
#define SPRINTZ_LAYOUT_LOWDIM 2
int64_t sprintz_mi355x_compress_layout(int codec, int elem_bytes, const void* src, uint32_t len, void* dest, uint16_t ndims,
                                       int write_size, int layout);
int64_t sprintz_mi355x_decompress_layout(int codec, int elem_bytes, const void* src, void* dest, int layout);

/* ------------------------------------------------------------------------
 * (2) Batched device API (device pointers, asynchronous on `hip_stream`).
 * ---------------------------------------------------------------------- */

/* Worst-case compressed bytes of one chunk (a multiple of 128: slots of this stride start on 128-byte lines, and the
 * encoders flush whole lines). */
size_t sprintz_mi355x_compress_bound(int elem_bytes, uint32_t chunk_len, uint16_t ndims);

/* Number of chunks total_len splits into. */
uint64_t sprintz_mi355x_num_chunks(uint64_t total_len, uint32_t chunk_len);

/* Compress: d_src holds total_len elements, row-major [row][ndims]; chunk c
 * covers elements [c*chunk_len, min((c+1)*chunk_len, total_len)) and is
 * compressed exactly as sprintz_compress_<codec>_<w>b(src+c*chunk_len, n,
 * dest, ndims, true) would.  Chunk c's stream is written at
 * d_slots + c*slot_stride (slot_stride >= compress_bound, multiple of 16,
 * d_slots 16-byte aligned); d_sizes[c] receives its exact byte length and
 * d_rets[c] (optional, may be NULL) the reference's element-count return.
 * d_src must be readable for SPRINTZ_MI355X_READ_SLACK bytes past its end.
 * More than 2 047 columns: d_slots (and the decoder's d_out) must be ordinary DEVICE memory (hipMalloc) -- those
 * kernels build the stream with device-scope atomics on the slot and read their own output back; they also take
 * nchunks * ndims * 4 bytes of stream-ordered scratch (hipMallocAsync) per FIRE launch, so such a launch cannot be
 * captured into a graph.  From 4 096 columns on a shape whose verbatim tail can exceed the stream header's 16-bit
 * remaining_len (format.h:40; a chunk of whole blocks: 16 * ndims elements, else 8 * ndims + the ragged rest, or the
 * whole chunk when it is shorter than a group) is REFUSED with SPRINTZ_E_UNSUPPORTED by every batched entry point: the
 * reference's single call writes such a stream truncated (it decodes to a prefix), which the drop-in single-call
 * symbols reproduce and a batch must not. */
int sprintz_mi355x_compress_batch(int codec, int elem_bytes,
                                  const void* d_src, uint64_t total_len, uint32_t chunk_len, uint16_t ndims,
                                  void* d_slots, size_t slot_stride,
                                  uint32_t* d_sizes, int64_t* d_rets, void* hip_stream);

/* Compact slot-strided chunk streams into one dense buffer: exclusive scan of
 * sizes -> d_offsets[0..nchunks] (d_offsets[nchunks] = total bytes), then a
 * coalesced copy.  `align` (power of two, 1..16) rounds every chunk start up;
 * align=1 gives the byte-dense concatenation.  d_dense capacity must be
 * >= sum(round_up(size, align)) + SPRINTZ_MI355X_READ_SLACK.
 * d_scan_tmp: scratch of sprintz_mi355x_compact_tmp_bytes(nchunks) bytes. */
size_t sprintz_mi355x_compact_tmp_bytes(uint64_t nchunks);
int sprintz_mi355x_compact(const void* d_slots, size_t slot_stride, const uint32_t* d_sizes,
                           uint64_t nchunks, uint32_t align,
                           void* d_dense, uint64_t* d_offsets, void* d_scan_tmp, void* hip_stream);

/* Compress straight into the dense container: sprintz_mi355x_compress_batch followed by
 * sprintz_mi355x_compact(align = 16), with the same results in d_sizes / d_rets / d_dense / d_offsets -- but for the
 * shapes the fast encoder takes (general layout, ndims <= 64, 16-byte aligned blocks) in ONE launch: every workgroup
 * finds its place in the container by a chained scan over workgroups and copies its own chunks slot -> container before
 * it exits (csrc/compact_tail.h).  d_slots is still needed (the streams pass
 * through it); d_tmp: sprintz_mi355x_compress_dense_tmp_bytes(nchunks) bytes, 8-byte aligned; d_dense: 16-byte aligned,
 * capacity as for sprintz_mi355x_compact.  Other shapes run the two launches internally. */
size_t sprintz_mi355x_compress_dense_tmp_bytes(uint64_t nchunks);
int sprintz_mi355x_compress_batch_dense(int codec, int elem_bytes,
                                        const void* d_src, uint64_t total_len, uint32_t chunk_len, uint16_t ndims,
                                        void* d_slots, size_t slot_stride, uint32_t* d_sizes, int64_t* d_rets,
                                        void* d_dense, uint64_t* d_offsets, void* d_tmp, void* hip_stream);

/* Decompress: chunk c's stream occupies [d_offsets[c], d_offsets[c+1]) of d_comp
 * (any byte alignment; d_offsets has nchunks+1 entries, the last one being the
 * end of the last stream -- the layout sprintz_mi355x_compact produces; padding
 * between streams is allowed) and is decoded exactly as
 * sprintz_decompress_<codec>_<w>b would, to d_out + c*chunk_len elements.  d_rets[c] (optional) receives the
 * element count decoded (the reference's return value), or a negative
 * SPRINTZ_E_* if the stream header's ndims differs from `ndims`.
 * d_comp must be readable for SPRINTZ_MI355X_READ_SLACK bytes past the last
 * stream byte. */
int sprintz_mi355x_decompress_batch(int codec, int elem_bytes,
                                    const void* d_comp, const uint64_t* d_offsets, uint64_t nchunks,
                                    uint32_t chunk_len, uint16_t ndims,
                                    void* d_out, int64_t* d_rets, void* hip_stream);

/* ------------------------------------------------------------------------
 * (3) Optional Huffman stage over the container (bytes as symbols, after
 * bit-packing -- what the paper does with Huff0, communicate/ubicomp/
 * method.tex:293-297).  The reference tree contains NO Huffman coder
 * (SURVEY.md 8c), so this container format is this library's own (specified in
 * oracle/huf_oracle.c: 64-chunk segments share a code table, 4 byte-aligned
 * sub-streams per chunk, 11-bit code limit) and its parity is unpinned; the
 * Sprintz streams it wraps remain bit-exact with the reference.
 *   d_tables : ceil(nchunks/64) * 128 bytes of code-length tables
 *   d_huf    : capacity >= sprintz_mi355x_huf_bound(sum of d_sizes, nchunks)
 *   d_tmp    : sprintz_mi355x_huf_tmp_bytes(nchunks) bytes of scratch
 * ---------------------------------------------------------------------- */
size_t sprintz_mi355x_huf_tmp_bytes(uint64_t nchunks);
size_t sprintz_mi355x_huf_bound(uint64_t total_stream_bytes, uint64_t nchunks);
/* container (d_dense, d_offsets[nchunks+1], d_sizes[nchunks] = exact stream bytes)
 * -> Huffman records at d_huf + d_huf_offsets[c] (d_huf_offsets[nchunks] = total) */
int sprintz_mi355x_huf_compress_batch(const void* d_dense, const uint64_t* d_offsets, const uint32_t* d_sizes,
                                      uint64_t nchunks, void* d_huf, uint64_t* d_huf_offsets, void* d_tables,
                                      void* d_tmp, void* hip_stream);
/* inverse: rebuilds the container with chunk starts rounded up to `align`
 * (fills d_offsets[nchunks+1] and d_sizes[nchunks]); feed it to
 * sprintz_mi355x_decompress_batch.  d_dense holds dense_capacity bytes; a
 * record that is damaged (does not fit its slot of the container, or whose
 * output would not fit d_dense) decodes to nothing and gets
 * d_rets[c] = SPRINTZ_E_CORRUPT (d_rets optional; otherwise the chunk's byte
 * count) -- damaged input never makes the kernels read or write out of bounds. */
int sprintz_mi355x_huf_decompress_batch(const void* d_huf, const uint64_t* d_huf_offsets, const void* d_tables,
                                        uint64_t nchunks, uint32_t align, void* d_dense, uint64_t dense_capacity,
                                        uint64_t* d_offsets, uint32_t* d_sizes, int64_t* d_rets, void* d_tmp,
                                        void* hip_stream);

/* ------------------------------------------------------------------------
 * Column-major matrices (BASELINE.json config 5: "uint16 colmajor, 32
 * variables").  The matrix X[nrows][ndims] lives in HBM column by column:
 * element (r, d) at base[d * col_stride + r].  Chunk c holds rows
 * [c*rows_per_chunk, min((c+1)*rows_per_chunk, nrows)) and its stream is
 * byte-for-byte what the reference's compress() produces for the row-major
 * flattening of those rows (len = rows * ndims) -- the transposition happens
 * in the kernels' addressing (a lane owns a column, so its 8 samples of a
 * block are one contiguous 16-byte piece), not in memory.
 *   compress:   col_stride >= nrows
 *   decompress: col_stride >= nchunks * rows_per_chunk (ragged last chunk:
 *               rows past nrows are not written by a valid stream)
 * Containers are interchangeable with the row-major entry points.
 * ---------------------------------------------------------------------- */
int sprintz_mi355x_compress_batch_colmajor(int codec, int elem_bytes, const void* d_src, uint64_t nrows, uint64_t col_stride,
                                           uint32_t rows_per_chunk, uint16_t ndims, void* d_slots, size_t slot_stride,
                                           uint32_t* d_sizes, int64_t* d_rets, void* hip_stream);
/* the column-major write path in one call: streams + the 16-byte aligned container + offsets[nchunks + 1], as
 * sprintz_mi355x_compress_batch_dense does for row-major data (same d_tmp size, same fallback to encode + scan + copy for
 * the shapes whose encoder carries no container tail) */
int sprintz_mi355x_compress_batch_colmajor_dense(int codec, int elem_bytes, const void* d_src, uint64_t nrows, uint64_t col_stride,
                                                 uint32_t rows_per_chunk, uint16_t ndims, void* d_slots, size_t slot_stride,
                                                 uint32_t* d_sizes, int64_t* d_rets, void* d_dense, uint64_t* d_offsets, void* d_tmp,
                                                 void* hip_stream);
int sprintz_mi355x_decompress_batch_colmajor(int codec, int elem_bytes, const void* d_comp, const uint64_t* d_offsets,
                                             uint64_t nchunks, uint32_t rows_per_chunk, uint16_t ndims, uint64_t col_stride,
                                             void* d_out, int64_t* d_rets, void* hip_stream);

/* ------------------------------------------------------------------------
 * Query on compressed data.  Replaces
 *   query_rowmajor_delta_rle_{8b,16b}(src, dest, const QueryParams&)  cpp/Compress/sprintz_delta.h:95-98
 *   query_rowmajor_xff_rle_{8b,16b}(src, dest, const QueryParams&)    cpp/Compress/sprintz_xff.h:90-93
 *   QueryParams{op, materialize}, QueryTypes::{NOOP,REDUCE_MAX,REDUCE_SUM}  cpp/Compress/query.hpp:23-29
 * The reduction is fused into the decode kernel: with materialize == 0 nothing
 * but the per-column results leaves the chip.
 *
 * Semantics.  The reference computes its reductions into a local object and
 * throws them away (sprintz_xff_rle_query.cpp:69-104: DUMMY_READ_QUERY_RESULT),
 * and its functors are unfinished (query.hpp:222 writes every stripe's maximum
 * to state[0]; query.hpp:84-87,120-123 shift the wrong way and drop columns
 * 8..15 of every 32), so there is no reference RESULT to be bit-exact with.
 * Defined here: for column c (element index mod ndims), over ALL decompressed
 * elements of the chunk including its verbatim tail --
 *   op 1 (REDUCE_MAX): the unsigned maximum;  op 2 (REDUCE_SUM): the sum of the
 *   unsigned values, in 64 bits.
 * The materialised output is bit-exact with decompress (which is what the
 * reference's own query tests assert, test/test_query.cpp:59-120,180-200).
 *
 *   op          : SPRINTZ_QUERY_NOOP / _MAX / _SUM
 *   materialize : 0 = do not write the decompressed data (d_out may be NULL)
 *   flags       : SPRINTZ_QUERY_GENERAL_LAYOUT = the stream uses the general
 *                 row-major layout whatever ndims is, as the reference's
 *                 *_rowmajor_*_rle_* functions do (sprintz.h's dispatch uses the
 *                 low-dim layout for ndims <= 4 @8b / <= 2 @16b; that is the default)
 *   d_partials  : [nchunks][ndims] uint64, per-chunk per-column results
 *   d_rets      : optional, as in decompress_batch
 * sprintz_mi355x_query_reduce folds the partials over the chunks:
 *   d_result[ndims] = max / sum over chunks.
 * ---------------------------------------------------------------------- */
#define SPRINTZ_QUERY_NOOP 0
#define SPRINTZ_QUERY_MAX 1
#define SPRINTZ_QUERY_SUM 2
#define SPRINTZ_QUERY_GENERAL_LAYOUT 1u
int sprintz_mi355x_query_batch(int codec, int elem_bytes, const void* d_comp, const uint64_t* d_offsets, uint64_t nchunks,
                               uint32_t chunk_len, uint16_t ndims, int op, int materialize, uint32_t flags, void* d_out,
                               uint64_t* d_partials, int64_t* d_rets, void* hip_stream);
int sprintz_mi355x_query_reduce(int op, const uint64_t* d_partials, uint64_t nchunks, uint16_t ndims, uint64_t* d_result,
                                void* hip_stream);
/* single-call forms over host buffers; result: ndims uint64 (may be NULL);
 * return value as decompress (elements), < 0 on error */
int64_t sprintz_mi355x_query_delta_8b(const int8_t* src, uint8_t* dest, int op, int materialize, uint32_t flags, uint64_t* result);
int64_t sprintz_mi355x_query_delta_16b(const int16_t* src, uint16_t* dest, int op, int materialize, uint32_t flags, uint64_t* result);
int64_t sprintz_mi355x_query_xff_8b(const int8_t* src, uint8_t* dest, int op, int materialize, uint32_t flags, uint64_t* result);
int64_t sprintz_mi355x_query_xff_16b(const int16_t* src, uint16_t* dest, int op, int materialize, uint32_t flags, uint64_t* result);

/* single-call forms of the non-RLE codecs over host buffers; replace
 *   compress_rowmajor_{8b,16b} / decompress_rowmajor_{8b,16b}                cpp/Compress/sprintz_delta.h:26-31,63-66
 *   compress_rowmajor_delta_{8b,16b} / decompress_rowmajor_delta_{8b,16b}    cpp/Compress/sprintz_delta.h:37-42,72-76
 *   compress8b_rowmajor_xff / decompress8b_rowmajor_xff                      cpp/Compress/sprintz_xff.h:28-31
 * codec: SPRINTZ_CODEC_DELTA_NORLE, SPRINTZ_CODEC_BITPACK_NORLE or SPRINTZ_CODEC_XFF_NORLE; return values in elements as above */
int64_t sprintz_mi355x_compress_norle(int codec, int elem_bytes, const void* src, uint32_t len, void* dest, uint16_t ndims);
int64_t sprintz_mi355x_decompress_norle(int codec, int elem_bytes, const void* src, void* dest);

/* ------------------------------------------------------------------------
 * Huff0 wire format, both directions.  The paper's entropy stage is Yann Collet's
 * Huff0 (communicate/ubicomp/method.tex:293-297), a third-party coder the
 * reference neither vendors nor pins; this entry point decodes GENUINE Huff0
 * blocks -- what HUF_compress of zstd 1.4.x / lzbench's huff0 writes: tree
 * description (4-bit or FSE-compressed weights), 3 x u16 jump table, 4 bit
 * streams -- one block per chunk, so that a container entropy-coded by that
 * library can be taken straight to sprintz_mi355x_decompress_batch.  Replaces
 * HUF_decompress(dst, dstSize, cSrc, cSrcSize) per chunk, with its conventions:
 * block bytes == decoded bytes means stored, 1 byte means one repeated byte.
 *   d_blocks + d_block_offsets[c] .. [c+1] : chunk c's block
 *   d_out + d_out_offsets[c] .. [c+1]      : where its bytes go (the sizes are
 *                                            the caller's, as with HUF_decompress)
 *   d_rets[c] (optional): decoded bytes, or SPRINTZ_E_CORRUPT for a damaged block
 *   (nothing is read or written outside the chunk's two ranges).
 * d_blocks must be 16-byte aligned and readable 16 bytes past its end.  Format restated in
 * oracle/huf0_oracle.c; kernel in sprintz_amd/csrc/huf0.hip.
 * ---------------------------------------------------------------------- */
int sprintz_mi355x_huf0_decompress_batch(const void* d_blocks, const uint64_t* d_block_offsets, uint64_t nchunks, void* d_out,
                                         const uint64_t* d_out_offsets, int64_t* d_rets, void* hip_stream);
/* The same with the caller's workspace (d_tmp: sprintz_mi355x_huf0_decode_tmp_bytes(nchunks) bytes, 16-byte aligned;
 * holds one 320-byte code descriptor per chunk and two flag arrays between the launches).  The form above allocates it
 * stream-ordered (hipMallocAsync) per call; a caller with a steady batch size passes its own.  A block's
 * streams are limited to 128 MiB each (HUF_compress never writes a block above 128 KB). */
size_t sprintz_mi355x_huf0_decode_tmp_bytes(uint64_t nchunks);
int sprintz_mi355x_huf0_decompress_batch_ws(const void* d_blocks, const uint64_t* d_block_offsets, uint64_t nchunks, void* d_out,
                                            const uint64_t* d_out_offsets, int64_t* d_rets, void* d_tmp, void* hip_stream);
/* The same with a HINT: max_block_bytes = an upper bound of the batch's block sizes (0 = unknown).  Small batches
 * (SPRINTZ_OPT_HUF0_SYNC_CHUNKS) keep a chunk's whole block in LDS; the hint sizes that image (up to 16 KB a block; without one 4 KB).
 * A block above the image -- or above a wrong hint -- is still decoded, from global memory: the hint is never trusted for safety. */
int sprintz_mi355x_huf0_decompress_batch_hint(const void* d_blocks, const uint64_t* d_block_offsets, uint64_t nchunks, void* d_out,
                                              const uint64_t* d_out_offsets, int64_t* d_rets, void* d_tmp, uint32_t max_block_bytes,
                                              void* hip_stream);
/* The other direction: container (d_dense, d_offsets[nchunks+1], d_sizes[nchunks] = exact stream bytes, as
 * sprintz_mi355x_compact leaves them) -> one Huff0 block per chunk, byte-dense at d_blocks +
 * d_block_offsets[c] (d_block_offsets[nchunks] = total), each one a block HUF_decompress(dst, size of
 * the chunk, block, size of the block) decodes -- the write side of lzbench's huff0 stage.  The
 * format leaves the encoder free; what is written is specified by oracle/huf0_oracle.c
 * (oracle_huf0_compress_batch: one code table per 64 chunks repeated in every block, FSE-coded or 4-bit
 * weights, stored / one-byte blocks under HUF_decompress's conventions) and the kernels are
 * byte-exact with it.  d_blocks: sprintz_mi355x_huf0_bound(sum of sizes, nchunks) bytes; d_tmp:
 * sprintz_mi355x_huf0_tmp_bytes(nchunks). */
size_t sprintz_mi355x_huf0_tmp_bytes(uint64_t nchunks);
size_t sprintz_mi355x_huf0_bound(uint64_t total_stream_bytes, uint64_t nchunks);
int sprintz_mi355x_huf0_compress_batch(const void* d_dense, const uint64_t* d_offsets, const uint32_t* d_sizes, uint64_t nchunks,
                                       void* d_blocks, uint64_t* d_block_offsets, void* d_tmp, void* hip_stream);

/* ------------------------------------------------------------------------
 * Stand-alone transforms.  Replace
 *   encode_delta_rowmajor_{8b,16b} / decode_delta_rowmajor_{8b,16b}              cpp/Compress/delta.h:17-24,53-60
 *   encode_doubledelta_rowmajor_{8b,16b} / decode_doubledelta_rowmajor_{8b,16b}  cpp/Compress/delta.h:36-43,63-68
 *   encode_xff_rowmajor_{8b,16b} / decode_xff_rowmajor_{8b,16b}                  cpp/Compress/predict.h:15-30
 * Per column (element index mod ndims), state starting at zero, arithmetic
 * wrapping at the element width: delta y[r] = x[r] - x[r-1]; double delta
 * y[r] = x[r] - 2 x[r-1] + x[r-2].  One call transforms ONE stream of any
 * length; the decode is a scan over its rows (transforms.hip): one pass -- a chained scan
 * over 128 KB tiles on persistent workgroups -- for 16-bit streams whose
 * rows are 1 .. 8 whole 16-byte pieces (or 2 / 4 / 8 bytes), two passes otherwise (env
 * SPRINTZ_MI355X_TRANSFORM_CHAIN: 0 = two passes always, n = one pass from n tiles on;
 * A/B runs, tests).  The scratch is zeroed by the call (hipMemsetAsync on the stream).
 * SPRINTZ_TRANSFORM_XFF is the FIRE forecaster with no packing (errors out), with
 * predict.cpp's own constants (not the codec's); only whole 8-row blocks that its
 * vector stores cannot spill past the end are forecast, the rest is plain delta
 * (predict.cpp:96-103, :266-273).  Its counters make the rows of one stream strictly
 * sequential in both directions: a lane per column, latency-bound (transforms.hip).
 *   kind        : SPRINTZ_TRANSFORM_DELTA / SPRINTZ_TRANSFORM_DOUBLEDELTA / SPRINTZ_TRANSFORM_XFF
 * Device forms: len elements in, len elements out, no header;
 *   d_tmp: sprintz_mi355x_transform_tmp_bytes(kind, elem_bytes, len, ndims).
 * Host forms: the reference's container -- a 6-byte header {u32 len; u16 ndims}
 * (format.h:65-86) before the transformed elements; encode returns len + header
 * length in elements (6 @8b, 3 @16b; delta.cpp:120), decode returns len.  decode
 * with raw_len == raw_ndims == 0 reads the header; otherwise src is headerless
 * (the reference's 4-argument form, delta.h:19).
 * ---------------------------------------------------------------------- */
#define SPRINTZ_TRANSFORM_DELTA 0
#define SPRINTZ_TRANSFORM_DOUBLEDELTA 1
#define SPRINTZ_TRANSFORM_XFF 2
size_t sprintz_mi355x_transform_tmp_bytes(int kind, int elem_bytes, uint64_t len, uint16_t ndims);
int sprintz_mi355x_transform_encode_device(int kind, int elem_bytes, const void* d_src, uint64_t len, uint16_t ndims, void* d_dest,
                                           void* hip_stream);
int sprintz_mi355x_transform_decode_device(int kind, int elem_bytes, const void* d_src, uint64_t len, uint16_t ndims, void* d_dest,
                                           void* d_tmp, void* hip_stream);
int64_t sprintz_mi355x_transform_encode(int kind, int elem_bytes, const void* src, uint32_t len, void* dest, uint16_t ndims,
                                        int write_size);
int64_t sprintz_mi355x_transform_decode(int kind, int elem_bytes, const void* src, void* dest, uint32_t raw_len, uint16_t raw_ndims);
const char* sprintz_mi355x_transform_last_error(void);

/* ------------------------------------------------------------------------
 * The reference's 2020 "online" coders for 1-D uint16 streams.  Replace
 *   dynamic_delta_pack_u16 / dynamic_delta_pack_u16_altloss / dynamic_delta_unpack_u16   cpp/Compress/online.hpp:412-424
 *   zigzag_pack_u16 / zigzag_unpack_u16                                                  cpp/Compress/online.hpp:431-438
 *   sprintzpack_pack_u16 / _zigzag / sprintzpack_unpack_u16 / _zigzag                    cpp/Compress/online.hpp:450-462
 * One call codes ONE stream; the containers ({u32 len} + payload, restated in oracle/online_oracle.c) are byte-exact
 * with the reference on every byte it writes (it leaves header padding unwritten; here those bytes are 0).
 * Return values are ELEMENTS like the reference's.  Device forms: d_src / d_dest 16-byte aligned,
 * d_dest of sprintz_mi355x_online_bound() bytes, d_tmp of sprintz_mi355x_online_tmp_bytes(); *d_ret (device) receives the
 * return value -- for unpack, SPRINTZ_E_CORRUPT if the container's length field differs from `len`.
 * The dynamic-delta decoder takes streams of at least 128 tiles (of 8 192 blocks: 16 MB) in one pass -- a chained scan of the blocks' affine
 * maps on persistent workgroups -- and shorter ones in three launches (env SPRINTZ_MI355X_ONLINE_CHAIN: 0 = three launches always,
 * n = one pass from n tiles on; A/B runs, tests).
 * ---------------------------------------------------------------------- */
#define SPRINTZ_ONLINE_DYNDELTA 0        /* dynamic delta / double delta, loss SumLogAbs */
#define SPRINTZ_ONLINE_DYNDELTA_ALT 1    /* ... loss MaxAbs (encoder only differs) */
#define SPRINTZ_ONLINE_ZIGZAG 2
#define SPRINTZ_ONLINE_PACK 3            /* sprintzpack: per-block width + bit-packing, no zigzag */
#define SPRINTZ_ONLINE_PACK_ZIGZAG 4
size_t sprintz_mi355x_online_bound(int kind, uint32_t len);
size_t sprintz_mi355x_online_tmp_bytes(int kind, uint32_t len);
int sprintz_mi355x_online_pack_device(int kind, const uint16_t* d_src, uint32_t len, void* d_dest, int64_t* d_ret, void* d_tmp, void* hip_stream);
int sprintz_mi355x_online_unpack_device(int kind, const void* d_src, uint32_t len, uint16_t* d_dest, int64_t* d_ret, void* d_tmp, void* hip_stream);
int64_t sprintz_mi355x_online_pack(int kind, const uint16_t* src, uint32_t len, int16_t* dest);     /* host buffers */
int64_t sprintz_mi355x_online_unpack(int kind, const int16_t* src, uint16_t* dest);                 /* host buffers; len from the header */

/* ------------------------------------------------------------------------
 * Multi-GPU: one process per GPU, rank r owns a contiguous chunk range, the data
 * path has NO collective.  The only exchange is one all-gather of 8 bytes per rank -- each rank's
 * compressed byte count -- from which every rank derives where its container starts in the
 * job-wide one.  It runs over RCCL (xGMI), in-stream behind sprintz_mi355x_compact:
 *   rank 0:      sprintz_mi355x_comm_unique_id(id)   -> ship the 128 bytes to the other ranks
 *                                                        (any bootstrap: torchrun's store, MPI, a file)
 *   every rank:  sprintz_mi355x_comm_init(id, rank, world, &comm)          (on its own HIP device)
 *   per batch:   sprintz_mi355x_compact(..., d_offsets, ...)               d_offsets[nchunks] = local bytes
 *                sprintz_mi355x_gather_layout(comm, &d_offsets[nchunks], d_all, stream)   ncclAllGather
 *                sprintz_mi355x_layout_bases(d_all, world, bases, stream)  host: bases[r] = global byte offset
 *                                                                           of rank r's container, bases[world] = total
 * RCCL is loaded at run time (dlopen by SONAME; inside a torch process that is the copy
 * torch.distributed's "nccl" backend uses); without it comm_* return SPRINTZ_E_UNSUPPORTED.
 * ---------------------------------------------------------------------- */
#define SPRINTZ_MI355X_COMM_ID_BYTES 128
int sprintz_mi355x_comm_unique_id(void* id_out);
int sprintz_mi355x_comm_init(const void* id, int rank, int world, void** comm_out);
int sprintz_mi355x_gather_layout(void* comm, const uint64_t* d_local_total, uint64_t* d_all, void* hip_stream);
int sprintz_mi355x_layout_bases(const uint64_t* d_all, int world, uint64_t* bases_out /* world + 1 */, void* hip_stream);
int sprintz_mi355x_comm_destroy(void* comm);

/* ------------------------------------------------------------------------
 * Host convenience: chunked codec over host buffers (what lzbench does per
 * block).  Stages through device memory; PCIe-inclusive by construction.
 * comp layout: chunk streams concatenated byte-dense; offsets[nchunks+1].
 * ---------------------------------------------------------------------- */
int64_t sprintz_mi355x_compress_chunked_host(int codec, int elem_bytes, const void* src, uint64_t total_len,
                                             uint32_t chunk_len, uint16_t ndims,
                                             void* comp, size_t comp_capacity, uint64_t* offsets);
int64_t sprintz_mi355x_decompress_chunked_host(int codec, int elem_bytes, const void* comp,
                                               const uint64_t* offsets, uint64_t nchunks,
                                               uint32_t chunk_len, uint16_t ndims, void* out);

#ifdef __cplusplus
}
#endif
#endif /* SPRINTZ_MI355X_H */

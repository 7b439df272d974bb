/*
 * sprintz_oracle.h -- CPU restatement of the Sprintz codec hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle for the MI355X HIP
 * implementation: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may link or call it.  The shipped library
 * (libsprintz_mi355x.so) never calls into this code.
 *
 * What it restates (reference = dblalock/sprintz, cpp/Compress):
 *   sprintz.h:16-32                      the 8 public entry points
 *   sprintz.cpp:34-50                    ndims dispatch (low-dim vs general)
 *   format.h:36-62                       8-byte stream header
 *   sprintz_xff_rle.cpp:61-555           FIRE + zigzag + nbits + row-major pack + RLE
 *   sprintz_delta_rle.cpp:55-404         same with delta predictor
 *   sprintz_{delta,xff}_lowdim.cpp       column-major payload for D<=4 (8b) / D<=2 (16b)
 *
 * Parity status: PINNED.  The restatement is checked byte-for-byte (stream
 * bytes and return values) against (a) golden vectors minted from the
 * compiled reference (tests/golden/, generator oracle/gen_golden.py) and
 * (b) the compiled reference itself (oracle/_ref/libsprintz_ref.so) whenever
 * /root/reference is present (tests/test_oracle_vs_ref.py).
 *
 * Plain C11, scalar, no intrinsics, clean under -fsanitize=address,undefined.
 */
#ifndef SPRINTZ_ORACLE_H
#define SPRINTZ_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same signatures and return values (in ELEMENTS) as sprintz.h:16-32, with
 * write_size fixed to true.  `nbytes_out` (may be NULL) additionally receives
 * the exact number of compressed BYTES written: for 16-bit streams the
 * reference's element-count return value floors an odd byte length
 * (sprintz_xff_rle.cpp:554). */
int64_t oracle_compress_delta_8b (const uint8_t*  src, uint32_t len, int8_t*  dest, uint16_t ndims, size_t* nbytes_out);
int64_t oracle_compress_xff_8b   (const uint8_t*  src, uint32_t len, int8_t*  dest, uint16_t ndims, size_t* nbytes_out);
int64_t oracle_compress_delta_16b(const uint16_t* src, uint32_t len, int16_t* dest, uint16_t ndims, size_t* nbytes_out);
int64_t oracle_compress_xff_16b  (const uint16_t* src, uint32_t len, int16_t* dest, uint16_t ndims, size_t* nbytes_out);

int64_t oracle_decompress_delta_8b (const int8_t*  src, uint8_t*  dest);
int64_t oracle_decompress_xff_8b   (const int8_t*  src, uint8_t*  dest);
int64_t oracle_decompress_delta_16b(const int16_t* src, uint16_t* dest);
int64_t oracle_decompress_xff_16b  (const int16_t* src, uint16_t* dest);

/* Generic entry points used by the test harness / CPU baseline.
 *   codec: 0 = delta, 1 = xff (FIRE);  elem_bytes: 1 or 2.
 * oracle_decompress_q additionally exposes `ref_rle16_quirk`: when non-zero
 * the 16-bit general-layout FIRE run replay reproduces the reference
 * decoder's coefficient handling at sprintz_xff_rle.cpp:894-901 (shift by 4
 * instead of 12, odd columns not repositioned), which is NOT the inverse of
 * the reference encoder; see DESIGN.md "Reference decoder quirk". */
int64_t oracle_compress  (int codec, int elem_bytes, const void* src, uint32_t len, void* dest, uint16_t ndims, size_t* nbytes_out);
/* as above with the reference's `write_size` flag (false = no 8-byte header) */
int64_t oracle_compress_ws(int codec, int elem_bytes, const void* src, uint32_t len, void* dest, uint16_t ndims, int write_size, size_t* nbytes_out);
int64_t oracle_decompress(int codec, int elem_bytes, const void* src, void* dest);
int64_t oracle_decompress_q(int codec, int elem_bytes, const void* src, void* dest, int ref_rle16_quirk);
/* additionally reports how many stream BYTES the framing spans (header + groups + tail) */
int64_t oracle_decompress_ex(int codec, int elem_bytes, const void* src, void* dest, int ref_rle16_quirk,
                             size_t* consumed_bytes);

/* Worst-case compressed size in bytes for `len` elements of `ndims` columns. */
/* the reference's *_rowmajor_*_rle_* family (general layout whatever ndims is) and the
 * semantic definition of query-on-compressed; see sprintz_oracle.c */
int64_t oracle_compress_rowmajor(int codec, int elem_bytes, const void* src, uint32_t len, void* dest, uint16_t ndims, size_t* nbytes_out);
int64_t oracle_decompress_rowmajor_ex(int codec, int elem_bytes, const void* src, void* dest, int ref_rle16_quirk, size_t* consumed_bytes);
int64_t oracle_compress_norle(int raw, int elem_bytes, const void* src, uint32_t len, void* dest, uint16_t ndims, size_t* nbytes_out);
int64_t oracle_decompress_norle(int raw, int elem_bytes, const void* src, void* dest, size_t* consumed_bytes);
int64_t oracle_query(int codec, int elem_bytes, const void* src, void* dest, int general, int op, uint64_t* result);

size_t oracle_compress_bound(int elem_bytes, uint32_t len, uint16_t ndims);

/* Chunked helpers (what lzbench does with its block-size option, README.md:58):
 * every chunk of `chunk_len` elements (last one shorter) is an independent
 * compress() call.  Compressed chunk c is written at dest + c*dest_stride
 * bytes and its byte length stored in sizes[c].  Returns total bytes. */
uint64_t oracle_compress_chunks(int codec, int elem_bytes, const void* src, uint64_t total_len,
                                uint32_t chunk_len, uint16_t ndims,
                                uint8_t* dest, size_t dest_stride, uint32_t* sizes);
/* Decode chunks laid out at comp + offsets[c]; chunk c decodes to
 * out + c*chunk_len elements.  Returns total elements decoded. */
uint64_t oracle_decompress_chunks(int codec, int elem_bytes, const uint8_t* comp,
                                  const uint64_t* offsets, uint64_t nchunks,
                                  uint32_t chunk_len, void* out);

#ifdef __cplusplus
}
#endif
#endif

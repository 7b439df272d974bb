"""Shared test plumbing: ctypes bindings for the oracle (our CPU restatement)
and, when it has been built, the compiled reference (oracle/_ref); plus the
input families of the reference's own test-suite.

Input families mirror cpp/Compress/test/compress_testing.hpp:
  "known" squares + simple patterns  (:251-303)
  zeros                               (:380-386)
  fuzz: random then successively /2   (:347-370)
  sparse: random with most values 0   (:409-423)
and the size list of test_codec (:452-463).
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libsprintz_ref.so")

CODECS = {"delta": 0, "xff": 1}
DTYPES = {1: np.uint8, 2: np.uint16}

# sizes used by the reference's test_codec (compress_testing.hpp:452-463)
REF_TEST_SIZES = [1, 2, 7, 8, 15, 16, 17, 31, 32, 33, 63, 64, 66, 71, 72, 73, 127, 128,
                  129, 135, 136, 137, 4096, 4113]


def _bind(lib, name, restype, argtypes):
    f = getattr(lib, name)
    f.restype = restype
    f.argtypes = argtypes
    return f


class Oracle:
    """Our scalar C restatement (oracle/sprintz_oracle.c)."""

    def __init__(self, path=ORACLE_SO):
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} missing: run `make -C oracle` or __graft_entry__.build()")
        self.lib = C.CDLL(path)
        self._compress = _bind(self.lib, "oracle_compress_ws", C.c_int64,
                               [C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint16,
                                C.c_int, C.POINTER(C.c_size_t)])
        self._decompress = _bind(self.lib, "oracle_decompress_q", C.c_int64,
                                 [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int])
        self._decompress_ex = _bind(self.lib, "oracle_decompress_ex", C.c_int64,
                                    [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_size_t)])
        self._bound = _bind(self.lib, "oracle_compress_bound", C.c_size_t,
                            [C.c_int, C.c_uint32, C.c_uint16])
        self._compress_chunks = _bind(self.lib, "oracle_compress_chunks", C.c_uint64,
                                      [C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint16,
                                       C.c_void_p, C.c_size_t, C.c_void_p])
        self._decompress_chunks = _bind(self.lib, "oracle_decompress_chunks", C.c_uint64,
                                        [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64,
                                         C.c_uint32, C.c_void_p])

    # ---- optional Huffman stage (oracle/huf_oracle.c; our own container, see its header)
    def _huf_bind(self):
        if not hasattr(self, "_huf_c"):
            self._huf_c = _bind(self.lib, "huf_oracle_compress", C.c_uint64,
                                [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p])
            self._huf_d = _bind(self.lib, "huf_oracle_decompress", C.c_uint64,
                                [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p])
            self._huf_len = _bind(self.lib, "huf_oracle_lengths", None, [C.c_void_p, C.c_void_p])

    def huf_lengths(self, counts):
        self._huf_bind()
        counts = np.ascontiguousarray(counts, dtype=np.uint32)
        lens = np.zeros(256, np.uint8)
        self._huf_len(counts.ctypes.data, lens.ctypes.data)
        return lens

    def huf_compress(self, dense, offsets, sizes):
        """-> (huf bytes, huf_offsets[nchunks+1], tables)"""
        self._huf_bind()
        dense = np.ascontiguousarray(dense, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        sizes = np.ascontiguousarray(sizes, dtype=np.uint32)
        n = len(sizes)
        out = np.zeros(int(sizes.astype(np.int64).sum()) + 16 * n + 64, np.uint8)
        ho = np.zeros(n + 1, np.uint64)
        tables = np.zeros(((n + 63) // 64) * 128, np.uint8)
        total = self._huf_c(dense.ctypes.data, offsets.ctypes.data, sizes.ctypes.data, n, out.ctypes.data,
                            ho.ctypes.data, tables.ctypes.data)
        return out[:total].copy(), ho, tables

    def huf_decompress(self, huf, huf_offsets, tables, dense_capacity, align=16):
        """-> (dense bytes, offsets[nchunks+1], sizes[nchunks])"""
        self._huf_bind()
        huf = np.ascontiguousarray(huf, dtype=np.uint8)
        huf = np.concatenate([huf, np.zeros(16, np.uint8)])
        huf_offsets = np.ascontiguousarray(huf_offsets, dtype=np.uint64)
        tables = np.ascontiguousarray(tables, dtype=np.uint8)
        n = len(huf_offsets) - 1
        dense = np.zeros(dense_capacity + 64, np.uint8)
        offs = np.zeros(n + 1, np.uint64)
        sizes = np.zeros(n, np.uint32)
        total = self._huf_d(huf.ctypes.data, huf_offsets.ctypes.data, tables.ctypes.data, n, align, dense.ctypes.data,
                            offs.ctypes.data, sizes.ctypes.data)
        return dense[:total].copy(), offs, sizes

    def huf0_compress(self, dense, offsets, sizes):
        """the Huff0-format writer's specification (oracle_huf0_compress_batch): -> (blocks, block_offsets[n+1])"""
        f = _bind(self.lib, "oracle_huf0_compress_batch", C.c_uint64, [C.c_void_p] * 3 + [C.c_uint64] + [C.c_void_p] * 2)
        dense = np.ascontiguousarray(dense, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        sizes = np.ascontiguousarray(sizes, dtype=np.uint32)
        n = len(sizes)
        out = np.zeros(int(sizes.astype(np.int64).sum()) + 64, np.uint8)
        oo = np.zeros(n + 1, np.uint64)
        total = f(dense.ctypes.data, offsets.ctypes.data, sizes.ctypes.data, n, out.ctypes.data, oo.ctypes.data)
        return out[:total].copy(), oo

    def huf0_table_log(self, block):
        """table log of a coded Huff0 block's tree description (oracle_huf0_read_stats)"""
        f = _bind(self.lib, "oracle_huf0_read_stats", C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t])
        block = np.ascontiguousarray(block, dtype=np.uint8)
        w = np.zeros(256, np.uint8)
        nsym, tl = C.c_uint(0), C.c_uint(0)
        r = f(w.ctypes.data, C.byref(nsym), C.byref(tl), block.ctypes.data, block.size)
        return int(tl.value) if r > 0 else -1

    def huf0_decompress(self, block, dst_size):
        """one genuine Huff0 block (oracle/huf0_oracle.c = HUF_decompress): -> (bytes, return value)"""
        f = _bind(self.lib, "oracle_huf0_decompress", C.c_int64, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t])
        block = np.ascontiguousarray(block, dtype=np.uint8)
        dst = np.zeros(max(dst_size, 1), np.uint8)
        ret = f(dst.ctypes.data, dst_size, block.ctypes.data, block.size)
        return dst[:dst_size], int(ret)

    def bound(self, esz, n, ndims):
        return int(self._bound(esz, n, ndims))

    def compress(self, codec, data, ndims, write_size=True):
        """-> (stream bytes as np.uint8, return value in elements)"""
        data = np.ascontiguousarray(data)
        esz = data.dtype.itemsize
        n = data.size
        out = np.full(self.bound(esz, n, ndims) + 64, 0xAB, dtype=np.uint8)
        nb = C.c_size_t(0)
        ret = self._compress(CODECS[codec], esz, data.ctypes.data, n, out.ctypes.data, ndims,
                             int(write_size), C.byref(nb))
        return out[:nb.value].copy(), int(ret)

    def decompress(self, codec, stream, esz, capacity, quirk=0):
        """-> (decoded elements as np array, return value)"""
        stream = np.ascontiguousarray(stream, dtype=np.uint8)
        padded = np.concatenate([stream, np.zeros(64, np.uint8)])
        out = np.full(capacity + 64, 0xCD, dtype=DTYPES[esz])
        ret = self._decompress(CODECS[codec], esz, padded.ctypes.data, out.ctypes.data, quirk)
        return out[:max(int(ret), 0)].copy(), int(ret)

    def stream_nbytes(self, codec, stream, esz, capacity):
        """byte length implied by the stream's own framing (buffer may be longer)"""
        stream = np.ascontiguousarray(stream, dtype=np.uint8)
        out = np.zeros(capacity + 64, dtype=DTYPES[esz])
        used = C.c_size_t(0)
        self._decompress_ex(CODECS[codec], esz, stream.ctypes.data, out.ctypes.data, 0, C.byref(used))
        return int(used.value)

    # ---- the *_rowmajor_*_rle_* family (general layout for every ndims) and the query semantics
    def compress_rowmajor(self, codec, data, ndims):
        f = _bind(self.lib, "oracle_compress_rowmajor", C.c_int64,
                  [C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint16, C.POINTER(C.c_size_t)])
        data = np.ascontiguousarray(data)
        esz = data.dtype.itemsize
        out = np.full(self.bound(esz, data.size, ndims) + 64, 0xAB, dtype=np.uint8)
        nb = C.c_size_t(0)
        ret = f(CODECS[codec], esz, data.ctypes.data, data.size, out.ctypes.data, ndims, C.byref(nb))
        return out[:nb.value].copy(), int(ret)

    def query(self, codec, stream, esz, capacity, op, general=False):
        """-> (decompressed data, per-column result uint64[ndims]); op: 1 max, 2 sum"""
        f = _bind(self.lib, "oracle_query", C.c_int64,
                  [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p])
        stream = np.ascontiguousarray(stream, dtype=np.uint8)
        padded = np.concatenate([stream, np.zeros(64, np.uint8)])
        ndims = int(stream[6]) | (int(stream[7]) << 8)
        out = np.zeros(capacity + 64, dtype=DTYPES[esz])
        res = np.zeros(max(ndims, 1), np.uint64)
        ret = f(CODECS[codec], esz, padded.ctypes.data, out.ctypes.data, int(general), op, res.ctypes.data)
        return out[:max(int(ret), 0)].copy(), res[:ndims]

    # ---- non-RLE codecs (sprintz_delta.cpp:64-1391): raw = 1 bit-packing only, raw = 0 delta + bit-packing
    def compress_norle(self, raw, data, ndims):
        f = _bind(self.lib, "oracle_compress_norle", C.c_int64,
                  [C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint16, C.POINTER(C.c_size_t)])
        data = np.ascontiguousarray(data)
        esz = data.dtype.itemsize
        out = np.full(data.size * esz * 2 + 6 + 64 * max(ndims, 1) + 256, 0xAB, np.uint8)
        nb = C.c_size_t(0)
        ret = f(int(raw), esz, data.ctypes.data, data.size, out.ctypes.data, ndims, C.byref(nb))
        return out[:nb.value].copy(), int(ret)

    def decompress_norle(self, raw, stream, esz):
        f = _bind(self.lib, "oracle_decompress_norle", C.c_int64, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p])
        stream = np.concatenate([np.ascontiguousarray(stream, dtype=np.uint8), np.zeros(64, np.uint8)])
        n = int(np.frombuffer(stream[:4].tobytes(), np.uint32)[0])
        out = np.zeros(n + 64, DTYPES[esz])
        ret = f(int(raw), esz, stream.ctypes.data, out.ctypes.data, None)
        return out[:n].copy(), int(ret)

    # ---- stand-alone transforms (oracle/transforms_oracle.c): kind 0 delta, 1 double delta
    def transform_encode(self, kind, data, ndims, write_size=True):
        """-> (container bytes, return value)"""
        f = _bind(self.lib, "oracle_transform_encode", C.c_uint32,
                  [C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint16, C.c_int])
        data = np.ascontiguousarray(data)
        esz = data.dtype.itemsize
        out = np.full(data.size * esz + 6 + 64, 0xAB, np.uint8)
        ret = f(kind, esz, data.ctypes.data, data.size, out.ctypes.data, ndims, int(write_size))
        return out[: data.size * esz + (6 if write_size else 0)].copy(), int(ret)

    def transform_decode(self, kind, container, esz):
        f = _bind(self.lib, "oracle_transform_decode", C.c_uint32, [C.c_int, C.c_int, C.c_void_p, C.c_void_p])
        container = np.ascontiguousarray(container, dtype=np.uint8)
        n = int(np.frombuffer(container[:4].tobytes(), np.uint32)[0])
        out = np.zeros(n + 64, DTYPES[esz])
        ret = f(kind, esz, container.ctypes.data, out.ctypes.data)
        return out[:n].copy(), int(ret)

    def compress_chunks(self, codec, data, chunk_len, ndims):
        """-> (list of per-chunk streams)"""
        data = np.ascontiguousarray(data)
        esz = data.dtype.itemsize
        n = data.size
        nchunks = (n + chunk_len - 1) // chunk_len
        stride = self.bound(esz, chunk_len, ndims)
        dest = np.zeros(nchunks * stride + 64, np.uint8)
        sizes = np.zeros(nchunks, np.uint32)
        self._compress_chunks(CODECS[codec], esz, data.ctypes.data, n, chunk_len, ndims,
                              dest.ctypes.data, stride, sizes.ctypes.data)
        return [dest[c * stride:c * stride + int(sizes[c])].copy() for c in range(nchunks)]

    def compress_chunks_mt(self, codec, data, chunk_len, ndims, threads=None):
        """every chunk of a large batch on the host's cores (ctypes releases the GIL: plain Python threads, one contiguous
        chunk range each) -> (dest, stride, sizes): chunk c's stream is dest[c * stride : c * stride + sizes[c]]"""
        from concurrent.futures import ThreadPoolExecutor
        data = np.ascontiguousarray(data)
        esz = data.dtype.itemsize
        n = data.size
        nchunks = (n + chunk_len - 1) // chunk_len
        stride = self.bound(esz, chunk_len, ndims)
        dest = np.zeros(nchunks * stride + 64, np.uint8)
        sizes = np.zeros(nchunks, np.uint32)
        if threads is None:
            threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        threads = max(1, min(threads, nchunks))
        edges = [nchunks * t // threads for t in range(threads + 1)]

        def work(t):
            c0, c1 = edges[t], edges[t + 1]
            if c1 > c0:
                e0, e1 = c0 * chunk_len, min(n, c1 * chunk_len)
                self._compress_chunks(CODECS[codec], esz, data.ctypes.data + e0 * esz, e1 - e0, chunk_len, ndims,
                                      dest.ctypes.data + c0 * stride, stride, sizes.ctypes.data + 4 * c0)
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(work, range(threads)))
        return dest, stride, sizes

    def decompress_chunks(self, codec, comp, offsets, esz, chunk_len, total_len):
        comp = np.ascontiguousarray(comp, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        out = np.zeros(total_len + 64, DTYPES[esz])
        self._decompress_chunks(CODECS[codec], esz, comp.ctypes.data, offsets.ctypes.data,
                                len(offsets), chunk_len, out.ctypes.data)
        return out[:total_len]


class Reference:
    """The compiled reference (oracle/_ref/libsprintz_ref.so); only exists where
    /root/reference was available at build time (it does travel to the GPU box)."""

    def __init__(self, path=REF_SO):
        self.lib = C.CDLL(path)
        self._compress = _bind(self.lib, "ref_compress", C.c_int64,
                               [C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint16, C.c_int])
        self._decompress = _bind(self.lib, "ref_decompress", C.c_int64,
                                 [C.c_int, C.c_int, C.c_void_p, C.c_void_p])
        self._compress_chunks = _bind(self.lib, "ref_compress_chunks", C.c_uint64,
                                      [C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint16,
                                       C.c_void_p, C.c_size_t, C.c_void_p])
        self._decompress_chunks = _bind(self.lib, "ref_decompress_chunks", C.c_uint64,
                                        [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64,
                                         C.c_uint32, C.c_void_p])

    @staticmethod
    def available():
        return os.path.exists(REF_SO)

    def has_query(self):
        return hasattr(self.lib, "ref_query")

    def has_transforms(self):
        return hasattr(self.lib, "ref_transform_encode")

    def has_norle(self):
        return hasattr(self.lib, "ref_compress_norle")

    def compress_norle_raw(self, raw, data, ndims):
        """compress_rowmajor[_delta]_{8b,16b}: -> (whole poison-filled output buffer, return value in elements)"""
        f = _bind(self.lib, "ref_compress_norle", C.c_int64, [C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint16])
        data = np.ascontiguousarray(data)
        esz = data.dtype.itemsize
        src = np.concatenate([data.ravel(), np.zeros(256, data.dtype)])
        out = np.full(data.size * esz * 2 + 1024, 0xAB, np.uint8)
        ret = f(int(raw), esz, src.ctypes.data, data.size, out.ctypes.data, ndims)
        return out, int(ret)

    def decompress_norle(self, raw, stream, esz):
        f = _bind(self.lib, "ref_decompress_norle", C.c_int64, [C.c_int, C.c_int, C.c_void_p, C.c_void_p])
        stream = np.concatenate([np.ascontiguousarray(stream, dtype=np.uint8), np.zeros(512, np.uint8)])
        n = int(np.frombuffer(stream[:4].tobytes(), np.uint32)[0])
        out = np.zeros(n + 512, DTYPES[esz])
        ret = f(int(raw), esz, stream.ctypes.data, out.ctypes.data)
        return out[:n].copy(), int(ret)

    def transform_encode(self, kind, data, ndims):
        """encode_{delta,doubledelta}_rowmajor_{8b,16b}(write_size=true) -> (container bytes, return value)"""
        f = _bind(self.lib, "ref_transform_encode", C.c_uint32,
                  [C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint16, C.c_int])
        data = np.ascontiguousarray(data)
        esz = data.dtype.itemsize
        src = np.concatenate([data.ravel(), np.zeros(128, data.dtype)])           # the vector loop reads past the end
        out = np.full(data.size * esz + 6 + 512, 0xAB, np.uint8)
        ret = f(kind, esz, src.ctypes.data, data.size, out.ctypes.data, ndims, 1)
        return out[: data.size * esz + 6].copy(), int(ret)

    def transform_decode(self, kind, container, esz):
        f = _bind(self.lib, "ref_transform_decode", C.c_uint32, [C.c_int, C.c_int, C.c_void_p, C.c_void_p])
        container = np.concatenate([np.ascontiguousarray(container, dtype=np.uint8), np.zeros(512, np.uint8)])
        n = int(np.frombuffer(container[:4].tobytes(), np.uint32)[0])
        out = np.zeros(n + 512, DTYPES[esz])
        ret = f(kind, esz, container.ctypes.data, out.ctypes.data)
        return out[:n].copy(), int(ret)

    def compress_rowmajor_raw(self, codec, data, ndims):
        """compress_rowmajor_{delta,xff}_rle_{8b,16b} (sprintz_delta.h:49, sprintz_xff.h:45-55):
        -> (whole poison-filled output buffer, return value in elements)"""
        f = _bind(self.lib, "ref_compress_rowmajor", C.c_int64,
                  [C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint16])
        data = np.ascontiguousarray(data)
        esz = data.dtype.itemsize
        n = data.size
        src = np.concatenate([data.ravel(), np.zeros(64, data.dtype)])
        cap = (n * 3 // 2 + 64) * esz + 16 * ndims * esz + 256
        out = np.full(cap, 0xAB, dtype=np.uint8)
        ret = f(CODECS[codec], esz, src.ctypes.data, n, out.ctypes.data, ndims)
        return out, int(ret)

    def query(self, codec, stream, esz, capacity, op, materialize, ndims_hint=64):
        """query_rowmajor_*(src, dest, QueryParams{op, materialize}): -> (dest[:ret], ret)"""
        f = _bind(self.lib, "ref_query", C.c_int64, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int])
        stream = np.ascontiguousarray(stream, dtype=np.uint8)
        padded = np.concatenate([stream, np.zeros(256, np.uint8)])
        out = np.full(capacity + 64 + 4 * max(ndims_hint, 32), 0xCD, dtype=DTYPES[esz])
        ret = f(CODECS[codec], esz, padded.ctypes.data, out.ctypes.data, op, int(materialize))
        return out[:max(int(ret), 0)].copy(), int(ret)

    def compress_raw(self, codec, data, ndims, write_size=True):
        """-> (whole poison-filled output buffer, return value in elements).
        Buffer sized as the reference's tests do (compress_testing.hpp:145-147)
        plus slack for its 8-byte read-modify-write windows."""
        data = np.ascontiguousarray(data)
        esz = data.dtype.itemsize
        n = data.size
        # the reference reads 8-byte windows past the end of the input rows
        src = np.concatenate([data.ravel(), np.zeros(64, data.dtype)])
        cap = (n * 3 // 2 + 64) * esz + 16 * ndims * esz + 256
        out = np.full(cap, 0xAB, dtype=np.uint8)
        ret = self._compress(CODECS[codec], esz, src.ctypes.data, n, out.ctypes.data, ndims, int(write_size))
        return out, int(ret)

    def decompress(self, codec, stream, esz, capacity, ndims_hint=64):
        stream = np.ascontiguousarray(stream, dtype=np.uint8)
        padded = np.concatenate([stream, np.zeros(256, np.uint8)])
        out = np.full(capacity + 64 + 4 * max(ndims_hint, 32), 0xCD, dtype=DTYPES[esz])
        ret = self._decompress(CODECS[codec], esz, padded.ctypes.data, out.ctypes.data)
        return out[:max(int(ret), 0)].copy(), int(ret)


class Zstd:
    """The system libzstd's Huff0 (HUF_compress / HUF_decompress, zstd 1.4.x): the stand-in for the
    un-vendored coder of the paper.  Test infrastructure; absent -> tests skip."""

    def __init__(self):
        z = C.CDLL("libzstd.so.1")
        z.HUF_compress.restype = C.c_size_t
        z.HUF_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        z.HUF_decompress.restype = C.c_size_t
        z.HUF_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        z.HUF_compressBound.restype = C.c_size_t
        z.HUF_compressBound.argtypes = [C.c_size_t]
        z.HUF_isError.restype = C.c_uint
        z.HUF_isError.argtypes = [C.c_size_t]
        z.ZSTD_versionNumber.restype = C.c_uint
        self.z = z
        self.version = int(z.ZSTD_versionNumber())

    def huf_compress(self, data, table_log=None):
        """-> block bytes with HUF_decompress's conventions: the input itself when HUF_compress
        declines (returns 0), one byte when it is a single repeated symbol.  table_log: HUF_compress2's
        huffLog (the format's maximum is 12; HUF_compress itself asks for 11)"""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        if data.size == 0:
            return data.copy()
        out = np.zeros(self.z.HUF_compressBound(data.size) + 8, np.uint8)
        if table_log is None:
            r = self.z.HUF_compress(out.ctypes.data, out.size, data.ctypes.data, data.size)
        else:
            self.z.HUF_compress2.restype = C.c_size_t
            self.z.HUF_compress2.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint, C.c_uint]
            r = self.z.HUF_compress2(out.ctypes.data, out.size, data.ctypes.data, data.size, 255, table_log)
        if self.z.HUF_isError(r) or r == 0:
            return data.copy()
        return out[:r].copy()

    def huf_decompress(self, block, dst_size):
        block = np.ascontiguousarray(block, dtype=np.uint8)
        dst = np.zeros(dst_size, np.uint8)
        r = self.z.HUF_decompress(dst.ctypes.data, dst_size, block.ctypes.data, block.size)
        return dst, (-1 if self.z.HUF_isError(r) else int(r))


# ----------------------------------------------------------------- inputs

def gen_known(n, esz):
    """squares pattern: (i%16)^2 + ((i/16)%16)  (compress_testing.hpp:295-303)"""
    i = np.arange(n, dtype=np.int64)
    return (((i % 16) ** 2) + ((i // 16) % 16)).astype(DTYPES[esz])


def gen_patterns(n, esz):
    """the simple patterns of compress_testing.hpp:251-284 (+ a few more)"""
    dt = DTYPES[esz]
    i = np.arange(n, dtype=np.int64)
    top = (1 << (8 * esz)) - 1
    yield "zeros", np.zeros(n, dt)
    yield "ones", np.ones(n, dt)
    yield "max", np.full(n, top, dt)
    yield "iota", (i % (top + 1)).astype(dt)
    yield "iota_mod64", (i % 64).astype(dt)
    yield "alt_0_max", np.where(i % 2 == 0, 0, top).astype(dt)
    yield "mod3", (i % 3).astype(dt)
    yield "squares", gen_known(n, esz)


def gen_fuzz(rng, n, esz, shift):
    """random then successive /2 to shrink the range (compress_testing.hpp:347-370)"""
    top = 1 << (8 * esz)
    x = rng.integers(0, top, size=n, dtype=np.int64)
    return (x >> shift).astype(DTYPES[esz])


def gen_sparse(rng, n, esz, frac):
    """mostly zeros (compress_testing.hpp:409-423)"""
    top = 1 << (8 * esz)
    x = rng.integers(0, top, size=n, dtype=np.int64)
    keep = rng.random(n) < frac
    return np.where(keep, x, 0).astype(DTYPES[esz])


def gen_walk(rng, n, ndims, esz, step, flat_every=0):
    """per-column wrapping random walk, steps uniform in
    [-step, step]; optionally every `flat_every`-th 64-row span is held at
    constant slope to trigger RLE."""
    rows = (n + ndims - 1) // ndims
    steps = rng.integers(-step, step + 1, size=(rows, ndims), dtype=np.int64)
    if flat_every:
        span = (np.arange(rows) // 64) % flat_every == 0
        steps[span] = 0
    x = np.cumsum(steps, axis=0) + rng.integers(0, 1 << (8 * esz), size=(1, ndims))
    x = np.mod(x, 1 << (8 * esz)).astype(DTYPES[esz])
    return x.ravel()[:n]

"""Host-side mirror of the reference interface for the codec path.

Two layers, both thin over the C-ABI (include/sprintz_mi355x.h):

* ``sprintz_compress_delta_8b(src, len, dest, ndims, write_size=True)`` & co:
  the eight functions of the reference's cpp/Compress/sprintz.h:16-32 with the
  same names, argument order, units (ELEMENTS) and return values; ``src`` and
  ``dest`` are caller-owned numpy arrays, as the reference's are caller-owned
  C buffers.  One call == one chunk on the GPU: correct, not fast.

* ``ChunkedCodec``: the batched device API on torch tensors resident in HBM;
  this is what bench.py measures.  Chunk == independent compress() call
  (lzbench block, reference README.md:58).
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib

_NP = {1: np.uint8, 2: np.uint16}
_CODEC_ID = {"delta": _lib.CODEC_DELTA, "xff": _lib.CODEC_XFF,
             "delta_norle": _lib.CODEC_DELTA_NORLE, "bitpack": _lib.CODEC_BITPACK_NORLE,    # sprintz_delta.cpp:64-1391
             "xff_norle": _lib.CODEC_XFF_NORLE}                                             # sprintz_xff.cpp:35-626, 8-bit only


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _compress(codec, esz, src, length, dest, ndims, write_size):
    src = np.ascontiguousarray(src)
    if src.dtype.itemsize != esz or src.size < length:
        raise ValueError("src dtype/size does not match the call")
    if not dest.flags["C_CONTIGUOUS"] or not dest.flags["WRITEABLE"]:
        raise ValueError("dest must be a writable contiguous array")
    need = _lib.compress_bound(esz, length, ndims) if ndims else 8
    if dest.nbytes < min(need, (length * 3 // 2 + 64) * esz):
        raise ValueError("dest too small (reference callers allocate len*3/2+64 elements)")
    return int(_lib.compress[(codec, esz)](_np_ptr(src), length, _np_ptr(dest), ndims, int(bool(write_size))))


def _decompress(codec, esz, src, dest):
    src = np.ascontiguousarray(src)
    if not dest.flags["C_CONTIGUOUS"] or not dest.flags["WRITEABLE"]:
        raise ValueError("dest must be a writable contiguous array")
    return int(_lib.decompress[(codec, esz)](_np_ptr(src), _np_ptr(dest)))


# ---- the reference's eight entry points (sprintz.h:16-32)
def sprintz_compress_delta_8b(src, len, dest, ndims, write_size=True):  # noqa: A002 - reference's name
    return _compress("delta", 1, src, len, dest, ndims, write_size)


def sprintz_decompress_delta_8b(src, dest):
    return _decompress("delta", 1, src, dest)


def sprintz_compress_xff_8b(src, len, dest, ndims, write_size=True):  # noqa: A002
    return _compress("xff", 1, src, len, dest, ndims, write_size)


def sprintz_decompress_xff_8b(src, dest):
    return _decompress("xff", 1, src, dest)


def sprintz_compress_delta_16b(src, len, dest, ndims, write_size=True):  # noqa: A002
    return _compress("delta", 2, src, len, dest, ndims, write_size)


def sprintz_decompress_delta_16b(src, dest):
    return _decompress("delta", 2, src, dest)


def sprintz_compress_xff_16b(src, len, dest, ndims, write_size=True):  # noqa: A002
    return _compress("xff", 2, src, len, dest, ndims, write_size)


def sprintz_decompress_xff_16b(src, dest):
    return _decompress("xff", 2, src, dest)


def decompress_noheader(codec, esz, src, dest, ndims, ngroups, remaining_len):
    """5-argument kernel form for write_size=False streams (sprintz_xff.h:56-58)."""
    src = np.ascontiguousarray(src)
    return int(_lib.decompress_noheader(_CODEC_ID[codec], esz, _np_ptr(src), _np_ptr(dest), ndims, ngroups, remaining_len))


# ---- batched device API ------------------------------------------------------

@dataclass
class CompressedBatch:
    """Dense container: chunk c's stream is data[offsets[c]:offsets[c]+sizes[c]]
    (each stream bit-exact with the reference's output for that chunk)."""
    data: "torch.Tensor"       # uint8, device; readable READ_SLACK bytes past total
    offsets: "torch.Tensor"    # int64 [nchunks+1], device (offsets[-1] = total bytes incl. alignment padding)
    sizes: "torch.Tensor"      # int32 [nchunks], device: exact stream bytes
    nchunks: int
    total_len: int             # elements before compression
    chunk_len: int
    ndims: int

    def total_bytes(self):
        return int(self.offsets[-1].item())

    def stream_bytes(self):
        return int(self.sizes.to("cpu", dtype=__import__("torch").int64).sum().item())


class ChunkedCodec:
    """Batched Sprintz codec on one GPU.

    codec: "delta" | "xff"; elem_bytes: 1 | 2; ndims: columns; chunk_len:
    elements per independent chunk (10 KB of uint16 = 5120).
    """

    def __init__(self, codec, elem_bytes, ndims, chunk_len, device=None, align=16):
        import torch
        if codec not in _CODEC_ID:
            raise ValueError("codec must be 'delta', 'xff', 'delta_norle', 'bitpack' or 'xff_norle'")
        if elem_bytes not in (1, 2):
            raise ValueError("elem_bytes must be 1 or 2")
        if not torch.cuda.is_available():
            raise _lib.SprintzError(_lib.E_NO_DEVICE, "no HIP device visible to torch; there is no CPU fallback")
        self.torch = torch
        self.codec, self.esz, self.ndims, self.chunk_len, self.align = codec, elem_bytes, int(ndims), int(chunk_len), align
        dev = torch.device(device if device is not None else "cuda")
        if dev.type != "cuda":
            raise ValueError("device must be a cuda (HIP) device")
        # always an indexed device: "cuda" alone would never compare equal to a tensor's device
        self.device = torch.device("cuda", torch.cuda.current_device() if dev.index is None else dev.index)
        self.dtype = torch.uint8 if elem_bytes == 1 else torch.uint16
        self.slot_stride = int(_lib.compress_bound(elem_bytes, self.chunk_len, self.ndims))
        self._ws = {}

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def _on(self):
        """the C library launches on the CURRENT HIP device: make it this codec's for the call"""
        return self.torch.cuda.device(self.device)

    def workspace(self, nchunks):
        """slot buffer / sizes / scan scratch, cached per nchunks"""
        t = self.torch
        ws = self._ws.get(nchunks)
        if ws is None:
            ws = dict(
                slots=t.empty(nchunks * self.slot_stride, dtype=t.uint8, device=self.device),
                sizes=t.empty(nchunks, dtype=t.int32, device=self.device),
                rets=t.empty(nchunks, dtype=t.int64, device=self.device),
                tmp=t.empty(int(_lib.compress_dense_tmp_bytes(nchunks)), dtype=t.uint8, device=self.device),
            )
            self._ws = {nchunks: ws}
        return ws

    def _padded_view(self, src):
        """device tensor readable READ_SLACK bytes past its end (copy only if needed)"""
        t = self.torch
        flat = src.reshape(-1)
        buf = t.empty(flat.numel() * self.esz + _lib.READ_SLACK, dtype=t.uint8, device=self.device)
        buf[:flat.numel() * self.esz] = flat.view(t.uint8)
        return buf

    def compress_to_slots(self, src_padded_u8, total_len, ws=None):
        """encode kernel only: src (uint8 view, padded) -> slot-strided streams + sizes"""
        nchunks = int(_lib.num_chunks(total_len, self.chunk_len))
        ws = ws or self.workspace(nchunks)
        with self._on():
            _lib.check(_lib.compress_batch(_CODEC_ID[self.codec], self.esz, src_padded_u8.data_ptr(), total_len,
                                           self.chunk_len, self.ndims, ws["slots"].data_ptr(), self.slot_stride,
                                           ws["sizes"].data_ptr(), ws["rets"].data_ptr(), self._stream()))
        return ws

    def compact(self, ws, nchunks, dense=None, offsets=None):
        t = self.torch
        if dense is None:
            dense = t.empty(nchunks * self.slot_stride + _lib.READ_SLACK, dtype=t.uint8, device=self.device)
        if offsets is None:
            offsets = t.empty(nchunks + 1, dtype=t.int64, device=self.device)
        with self._on():
            _lib.check(_lib.compact(ws["slots"].data_ptr(), self.slot_stride, ws["sizes"].data_ptr(), nchunks, self.align,
                                    dense.data_ptr(), offsets.data_ptr(), ws["tmp"].data_ptr(), self._stream()))
        return dense, offsets

    def compress_dense(self, src_padded_u8, total_len, ws=None, dense=None, offsets=None):
        """the whole write path in one call (what the bench times): src -> dense container + offsets.  For the shapes
        the fast encoder takes this is ONE launch (the container is built inside it, csrc/compact_tail.h); align 16 only."""
        t = self.torch
        if self.align != 16:
            raise ValueError("compress_dense builds the 16-byte aligned container; use compress_to_slots + compact for another alignment")
        nchunks = int(_lib.num_chunks(total_len, self.chunk_len))
        ws = ws or self.workspace(nchunks)
        if dense is None:
            dense = t.empty(nchunks * self.slot_stride + _lib.READ_SLACK, dtype=t.uint8, device=self.device)
        if offsets is None:
            offsets = t.empty(nchunks + 1, dtype=t.int64, device=self.device)
        with self._on():
            _lib.check(_lib.compress_batch_dense(_CODEC_ID[self.codec], self.esz, src_padded_u8.data_ptr(), total_len,
                                                 self.chunk_len, self.ndims, ws["slots"].data_ptr(), self.slot_stride,
                                                 ws["sizes"].data_ptr(), ws["rets"].data_ptr(), dense.data_ptr(),
                                                 offsets.data_ptr(), ws["tmp"].data_ptr(), self._stream()))
        return ws, dense, offsets

    def compress(self, src):
        """src: device tensor of dtype uint8/uint16, any shape, row-major [.., ndims]."""
        t = self.torch
        if src.dtype.itemsize != self.esz or src.dtype.is_floating_point or src.device != self.device:
            raise ValueError(f"src must be a {self.esz}-byte integer tensor on {self.device}")
        total_len = src.numel()
        nchunks = int(_lib.num_chunks(total_len, self.chunk_len))
        if self.align == 16:
            ws, dense, offsets = self.compress_dense(self._padded_view(src.contiguous()), total_len)
        else:
            ws = self.compress_to_slots(self._padded_view(src.contiguous()), total_len)
            dense, offsets = self.compact(ws, nchunks)
        total = int(offsets[-1].item())
        data = dense[: total + _lib.READ_SLACK].clone()
        return CompressedBatch(data, offsets, ws["sizes"].clone(), nchunks, total_len, self.chunk_len, self.ndims)

    def decompress(self, batch, out=None, rets=None):
        """-> device tensor of total_len elements (chunk c at c*chunk_len)."""
        t = self.torch
        if out is None:
            out = t.empty(batch.nchunks * self.chunk_len, dtype=self.dtype, device=self.device)
        self.decompress_into(batch.data, batch.offsets, batch.nchunks, out, rets)
        return out[: batch.total_len]

    def decompress_into(self, data, offsets, nchunks, out, rets=None):
        """decode kernel only (what the bench times)"""
        with self._on():
            _lib.check(_lib.decompress_batch(_CODEC_ID[self.codec], self.esz, data.data_ptr(), offsets.data_ptr(), nchunks,
                                             self.chunk_len, self.ndims, out.data_ptr(),
                                             rets.data_ptr() if rets is not None else None, self._stream()))


    # ---- column-major matrices (BASELINE config 5): cols is a [ndims, col_stride] tensor, variable d in row d
    def compress_colmajor(self, cols, nrows=None):
        """cols[d, r] = sample r of variable d.  Chunk = chunk_len/ndims rows of all variables; the
        streams are what the reference produces for the row-major flattening of those rows."""
        t = self.torch
        if cols.dim() != 2 or cols.shape[0] != self.ndims or cols.dtype.itemsize != self.esz or not cols.is_contiguous():
            raise ValueError("cols must be a contiguous [ndims, col_stride] tensor of the codec's element type")
        if self.chunk_len % self.ndims:
            raise ValueError("chunk_len must be a multiple of ndims for column-major data")
        col_stride = int(cols.shape[1])
        nrows = col_stride if nrows is None else int(nrows)
        rows_per_chunk = self.chunk_len // self.ndims
        nchunks = (nrows + rows_per_chunk - 1) // rows_per_chunk
        ws = self.workspace(nchunks)
        if self.align == 16:                                # encode + container in one call (one launch where the encoder carries the tail)
            dense = t.empty(nchunks * self.slot_stride + _lib.READ_SLACK, dtype=t.uint8, device=self.device)
            offsets = t.empty(nchunks + 1, dtype=t.int64, device=self.device)
            with self._on():
                _lib.check(_lib.compress_batch_colmajor_dense(_CODEC_ID[self.codec], self.esz, cols.data_ptr(), nrows, col_stride, rows_per_chunk,
                                                              self.ndims, ws["slots"].data_ptr(), self.slot_stride, ws["sizes"].data_ptr(),
                                                              ws["rets"].data_ptr(), dense.data_ptr(), offsets.data_ptr(), ws["tmp"].data_ptr(),
                                                              self._stream()))
        else:
            with self._on():
                _lib.check(_lib.compress_batch_colmajor(_CODEC_ID[self.codec], self.esz, cols.data_ptr(), nrows, col_stride, rows_per_chunk,
                                                        self.ndims, ws["slots"].data_ptr(), self.slot_stride, ws["sizes"].data_ptr(),
                                                        ws["rets"].data_ptr(), self._stream()))
            dense, offsets = self.compact(ws, nchunks)
        total = int(offsets[-1].item())
        return CompressedBatch(dense[: total + _lib.READ_SLACK].clone(), offsets, ws["sizes"].clone(), nchunks,
                               nrows * self.ndims, self.chunk_len, self.ndims)

    def decompress_colmajor(self, batch, out=None):
        """-> [ndims, nrows] tensor (out, if given: [ndims, col_stride >= nchunks*rows_per_chunk])"""
        t = self.torch
        rows_per_chunk = self.chunk_len // self.ndims
        nrows = batch.total_len // self.ndims
        if out is None:
            out = t.empty((self.ndims, batch.nchunks * rows_per_chunk), dtype=self.dtype, device=self.device)
        with self._on():
            _lib.check(_lib.decompress_batch_colmajor(_CODEC_ID[self.codec], self.esz, batch.data.data_ptr(), batch.offsets.data_ptr(),
                                                      batch.nchunks, rows_per_chunk, self.ndims, int(out.shape[1]), out.data_ptr(),
                                                      None, self._stream()))
        return out[:, :nrows]

    def query(self, batch, op, materialize=False, out=None, reduce=True):
        """Query on the compressed container (query.hpp:23-29): per-column max / sum fused into
        the decode; returns (result, out).  result: uint64 tensor [ndims] (reduce=True) or the
        per-chunk partials [nchunks, ndims]; out: the decompressed elements if materialize."""
        torch = self.torch
        n = batch.nchunks
        opid = {None: _lib.QUERY_NOOP, "noop": _lib.QUERY_NOOP, "max": _lib.QUERY_MAX, "sum": _lib.QUERY_SUM}[op]
        if materialize and out is None:
            out = torch.empty(n * self.chunk_len, dtype=self.dtype, device=self.device)
        partials = torch.empty((n, self.ndims), dtype=torch.int64, device=self.device) if opid else None
        with self._on():
            _lib.check(_lib.query_batch(_CODEC_ID[self.codec], self.esz, batch.data.data_ptr(), batch.offsets.data_ptr(), n,
                                        self.chunk_len, self.ndims, opid, int(bool(materialize)), 0,
                                        out.data_ptr() if materialize else None,
                                        partials.data_ptr() if opid else None, None, self._stream()))
        res = partials
        if opid and reduce:
            res = torch.empty(self.ndims, dtype=torch.int64, device=self.device)
            with self._on():
                _lib.check(_lib.query_reduce(opid, partials.data_ptr(), n, self.ndims, res.data_ptr(), self._stream()))
        return res, (out[: batch.total_len] if materialize else None)


# ---- query on compressed data, single call (the reference's names) -----------------------

class QueryTypes:                       # query.hpp:23-25
    NOOP, REDUCE_MAX, REDUCE_SUM = 0, 1, 2


@dataclass
class QueryParams:                      # query.hpp:27-30
    op: int = QueryTypes.NOOP
    materialize: bool = False


def _query(codec, esz, src, dest, qp, general):
    src = np.ascontiguousarray(src)
    ndims = int(src.view(np.uint8)[6]) | (int(src.view(np.uint8)[7]) << 8)
    result = np.zeros(max(ndims, 1), np.uint64)
    flags = _lib.QUERY_GENERAL_LAYOUT if general else 0
    ret = int(_lib.query[(codec, esz)](_np_ptr(src), _np_ptr(dest) if dest is not None else None, int(qp.op),
                                       int(bool(qp.materialize)), flags, _np_ptr(result)))
    return ret, result[:ndims]


def query_rowmajor_delta_rle_8b(src, dest, qp, general_layout=True):
    """sprintz_delta.h:95; returns (elements, per-column result).  general_layout=True is what the
    reference's *_rowmajor_*_rle_* streams use; pass False for streams made by sprintz_compress_*."""
    return _query("delta", 1, src, dest, qp, general_layout)


def query_rowmajor_delta_rle_16b(src, dest, qp, general_layout=True):
    return _query("delta", 2, src, dest, qp, general_layout)


def query_rowmajor_xff_rle_8b(src, dest, qp, general_layout=True):
    return _query("xff", 1, src, dest, qp, general_layout)


def query_rowmajor_xff_rle_16b(src, dest, qp, general_layout=True):
    return _query("xff", 2, src, dest, qp, general_layout)


# ---- non-RLE codecs, the reference's names (sprintz_delta.h:26-76) --------------------------------

def _c_norle(codec, esz, src, length, dest, ndims):
    src = np.ascontiguousarray(src)
    if src.dtype.itemsize != esz or src.size < length:
        raise ValueError("src dtype/size does not match the call")
    return int(_lib.compress_norle(codec, esz, _np_ptr(src), length, _np_ptr(dest), ndims))


def compress_rowmajor_8b(src, len, dest, ndims):  # noqa: A002
    return _c_norle(_lib.CODEC_BITPACK_NORLE, 1, src, len, dest, ndims)


def compress_rowmajor_16b(src, len, dest, ndims):  # noqa: A002
    return _c_norle(_lib.CODEC_BITPACK_NORLE, 2, src, len, dest, ndims)


def compress_rowmajor_delta_8b(src, len, dest, ndims):  # noqa: A002
    return _c_norle(_lib.CODEC_DELTA_NORLE, 1, src, len, dest, ndims)


def compress_rowmajor_delta_16b(src, len, dest, ndims):  # noqa: A002
    return _c_norle(_lib.CODEC_DELTA_NORLE, 2, src, len, dest, ndims)


def compress8b_rowmajor_xff(src, len, dest, ndims):  # noqa: A002 - sprintz_xff.h:28
    return _c_norle(_lib.CODEC_XFF_NORLE, 1, src, len, dest, ndims)


def decompress8b_rowmajor_xff(src, dest):
    return int(_lib.decompress_norle(_lib.CODEC_XFF_NORLE, 1, _np_ptr(np.ascontiguousarray(src)), _np_ptr(dest)))


def decompress_rowmajor_8b(src, dest):
    return int(_lib.decompress_norle(_lib.CODEC_BITPACK_NORLE, 1, _np_ptr(np.ascontiguousarray(src)), _np_ptr(dest)))


def decompress_rowmajor_16b(src, dest):
    return int(_lib.decompress_norle(_lib.CODEC_BITPACK_NORLE, 2, _np_ptr(np.ascontiguousarray(src)), _np_ptr(dest)))


def decompress_rowmajor_delta_8b(src, dest):
    return int(_lib.decompress_norle(_lib.CODEC_DELTA_NORLE, 1, _np_ptr(np.ascontiguousarray(src)), _np_ptr(dest)))


def decompress_rowmajor_delta_16b(src, dest):
    return int(_lib.decompress_norle(_lib.CODEC_DELTA_NORLE, 2, _np_ptr(np.ascontiguousarray(src)), _np_ptr(dest)))


# ---- stand-alone transforms (delta.h:17-68) -------------------------------------------------

def _enc_t(kind, esz, src, length, dest, ndims, write_size):
    src = np.ascontiguousarray(src)
    if src.dtype.itemsize != esz or src.size < length:
        raise ValueError("src dtype/size does not match the call")
    if dest.nbytes < length * esz + (6 if write_size else 0):
        raise ValueError("dest too small")
    return int(_lib.transform_encode(kind, esz, _np_ptr(src), length, _np_ptr(dest), ndims, int(bool(write_size))))


def _dec_t(kind, esz, src, dest, length=0, ndims=0):
    src = np.ascontiguousarray(src)
    return int(_lib.transform_decode(kind, esz, _np_ptr(src), _np_ptr(dest), length, ndims))


def encode_delta_rowmajor_8b(src, len, dest, ndims, write_size=True):  # noqa: A002 - reference's name (delta.h:17)
    return _enc_t(_lib.TRANSFORM_DELTA, 1, src, len, dest, ndims, write_size)


def encode_delta_rowmajor_16b(src, len, dest, ndims, write_size=True):  # noqa: A002 (delta.h:53)
    return _enc_t(_lib.TRANSFORM_DELTA, 2, src, len, dest, ndims, write_size)


def encode_doubledelta_rowmajor_8b(src, len, dest, ndims, write_size=True):  # noqa: A002 (delta.h:36)
    return _enc_t(_lib.TRANSFORM_DOUBLEDELTA, 1, src, len, dest, ndims, write_size)


def encode_doubledelta_rowmajor_16b(src, len, dest, ndims, write_size=True):  # noqa: A002 (delta.h:63)
    return _enc_t(_lib.TRANSFORM_DOUBLEDELTA, 2, src, len, dest, ndims, write_size)


def decode_delta_rowmajor_8b(src, dest, len=0, ndims=0):  # noqa: A002 - both reference forms (delta.h:19-24)
    return _dec_t(_lib.TRANSFORM_DELTA, 1, src, dest, len, ndims)


def decode_delta_rowmajor_16b(src, dest, len=0, ndims=0):  # noqa: A002
    return _dec_t(_lib.TRANSFORM_DELTA, 2, src, dest, len, ndims)


def decode_doubledelta_rowmajor_8b(src, dest, len=0, ndims=0):  # noqa: A002
    return _dec_t(_lib.TRANSFORM_DOUBLEDELTA, 1, src, dest, len, ndims)


def decode_doubledelta_rowmajor_16b(src, dest, len=0, ndims=0):  # noqa: A002
    return _dec_t(_lib.TRANSFORM_DOUBLEDELTA, 2, src, dest, len, ndims)


def encode_xff_rowmajor_8b(src, len, dest, ndims, write_size=True):  # noqa: A002 (predict.h:15)
    return _enc_t(_lib.TRANSFORM_XFF, 1, src, len, dest, ndims, write_size)


def encode_xff_rowmajor_16b(src, len, dest, ndims, write_size=True):  # noqa: A002 (predict.h:24)
    return _enc_t(_lib.TRANSFORM_XFF, 2, src, len, dest, ndims, write_size)


def decode_xff_rowmajor_8b(src, dest, len=0, ndims=0):  # noqa: A002 (predict.h:17-21)
    return _dec_t(_lib.TRANSFORM_XFF, 1, src, dest, len, ndims)


def decode_xff_rowmajor_16b(src, dest, len=0, ndims=0):  # noqa: A002 (predict.h:26-30)
    return _dec_t(_lib.TRANSFORM_XFF, 2, src, dest, len, ndims)


def transform_device(kind, x, ndims, inverse=False, out=None):
    """delta ("delta") / double delta ("doubledelta") / FIRE errors ("xff", predict.h) of one
    row-major stream resident in HBM (torch uint8/uint16 tensor); inverse=True undoes it (delta
    kinds: a multi-level scan over the rows; xff: a lane per column, sequential in the rows)."""
    import torch
    k = {"delta": _lib.TRANSFORM_DELTA, "doubledelta": _lib.TRANSFORM_DOUBLEDELTA, "xff": _lib.TRANSFORM_XFF}[kind]
    esz = x.dtype.itemsize
    x = x.contiguous().reshape(-1)
    if out is None:
        out = torch.empty_like(x)
    stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    if not inverse:
        with torch.cuda.device(x.device):
            _lib.check(_lib.transform_encode_device(k, esz, x.data_ptr(), x.numel(), ndims, out.data_ptr(), stream))
    else:
        tmp = torch.empty(int(_lib.transform_tmp_bytes(k, esz, x.numel(), ndims)), dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.transform_decode_device(k, esz, x.data_ptr(), x.numel(), ndims, out.data_ptr(), tmp.data_ptr(), stream))
    return out


# ---- optional Huffman stage (device) ------------------------------------------------

@dataclass
class HufBatch:
    """Huffman-coded container (format: oracle/huf_oracle.c; unpinned vs the reference)."""
    data: "torch.Tensor"       # uint8 records
    offsets: "torch.Tensor"    # int64 [nchunks+1]
    tables: "torch.Tensor"     # uint8 [ceil(nchunks/64)*128]
    nchunks: int
    total_len: int
    chunk_len: int
    ndims: int

    def total_bytes(self):
        return int(self.offsets[-1].item()) + int(self.tables.numel())


def huf_compress(batch):
    """CompressedBatch -> HufBatch (entropy-codes the chunk streams on the GPU)"""
    import torch
    dev = batch.data.device
    n = batch.nchunks
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    total = batch.stream_bytes()
    huf = torch.zeros(int(_lib.huf_bound(total, n)), dtype=torch.uint8, device=dev)   # record gaps (< 4 B) read as 0
    hoffs = torch.empty(n + 1, dtype=torch.int64, device=dev)
    tables = torch.empty(((n + 63) // 64) * 128, dtype=torch.uint8, device=dev)
    tmp = torch.empty(int(_lib.huf_tmp_bytes(n)), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.huf_compress_batch(batch.data.data_ptr(), batch.offsets.data_ptr(), batch.sizes.data_ptr(), n,
                                           huf.data_ptr(), hoffs.data_ptr(), tables.data_ptr(), tmp.data_ptr(), stream))
    end = int(hoffs[-1].item())
    return HufBatch(huf[: end + _lib.READ_SLACK].clone(), hoffs, tables, n, batch.total_len, batch.chunk_len, batch.ndims)


def huf_decompress(hb, dense_capacity, align=16, rets=None):
    """HufBatch -> CompressedBatch (the exact Sprintz container again).  rets: optional int64
    tensor [nchunks] receiving each chunk's byte count, or E_CORRUPT for a damaged record."""
    import torch
    dev = hb.data.device
    n = hb.nchunks
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    cap = dense_capacity + 16 * n
    dense = torch.zeros(cap + _lib.READ_SLACK, dtype=torch.uint8, device=dev)
    offs = torch.empty(n + 1, dtype=torch.int64, device=dev)
    sizes = torch.empty(n, dtype=torch.int32, device=dev)
    tmp = torch.empty(int(_lib.compact_tmp_bytes(n)) + 64, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.huf_decompress_batch(hb.data.data_ptr(), hb.offsets.data_ptr(), hb.tables.data_ptr(), n, align,
                                             dense.data_ptr(), cap, offs.data_ptr(), sizes.data_ptr(),
                                             rets.data_ptr() if rets is not None else None, tmp.data_ptr(), stream))
    return CompressedBatch(dense, offs, sizes, n, hb.total_len, hb.chunk_len, hb.ndims)


def huf0_compress(batch):
    """CompressedBatch -> (blocks uint8 tensor, block_offsets int64 [nchunks+1]): one genuine Huff0 block per
    chunk (readable by HUF_decompress / lzbench's huff0; format of the writer: oracle/huf0_oracle.c)"""
    import torch
    dev = batch.data.device
    n = batch.nchunks
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    total = int(batch.sizes.sum().item())
    blocks = torch.zeros(int(_lib.huf0_bound(total, n)), dtype=torch.uint8, device=dev)
    boffs = torch.empty(n + 1, dtype=torch.int64, device=dev)
    tmp = torch.empty(int(_lib.huf0_tmp_bytes(n)), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.huf0_compress_batch(batch.data.data_ptr(), batch.offsets.data_ptr(), batch.sizes.data_ptr(), n,
                                            blocks.data_ptr(), boffs.data_ptr(), tmp.data_ptr(), stream))
    return blocks, boffs


def huf0_decompress(blocks, block_offsets, out_offsets, rets=None, out=None, max_block_bytes=0):
    """Genuine Huff0 blocks (HUF_compress's output, one per chunk; torch uint8 tensor + int64 offsets
    [nchunks+1]) -> the bytes they encode, chunk c at out_offsets[c] (int64 [nchunks+1], device).
    `blocks` must be 16-byte aligned and carry 16 readable bytes past the last block.  Returns the uint8 output tensor
    (READ_SLACK bytes longer than out_offsets[-1], ready for ChunkedCodec.decompress_into).
    max_block_bytes: an upper bound of the blocks' sizes if the caller has one (a chunk's compress_bound, say) -- small
    batches size their LDS image of a block by it (sprintz_mi355x_huf0_decompress_batch_hint); 0 = unknown."""
    import torch
    dev = blocks.device
    n = block_offsets.numel() - 1
    if out is None:
        out = torch.zeros(int(out_offsets[-1].item()) + _lib.READ_SLACK, dtype=torch.uint8, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    with torch.cuda.device(dev):
        if max_block_bytes:
            tmp = torch.empty(int(_lib.huf0_decode_tmp_bytes(n)), dtype=torch.uint8, device=dev)
            _lib.check(_lib.huf0_decompress_batch_hint(blocks.data_ptr(), block_offsets.data_ptr(), n, out.data_ptr(), out_offsets.data_ptr(),
                                                       rets.data_ptr() if rets is not None else None, tmp.data_ptr(), int(max_block_bytes), stream))
        else:
            _lib.check(_lib.huf0_decompress_batch(blocks.data_ptr(), block_offsets.data_ptr(), n, out.data_ptr(), out_offsets.data_ptr(),
                                                  rets.data_ptr() if rets is not None else None, stream))
    return out


# ---- host convenience (lzbench-style, PCIe inclusive) ---------------------------

def compress_chunked(codec, data, ndims, chunk_len):
    """numpy in -> (stream bytes np.uint8, offsets np.uint64[nchunks+1])"""
    data = np.ascontiguousarray(data)
    esz = data.dtype.itemsize
    nchunks = int(_lib.num_chunks(data.size, chunk_len))
    cap = nchunks * int(_lib.compress_bound(esz, chunk_len, ndims)) + 64
    comp = np.empty(cap, np.uint8)
    offsets = np.zeros(nchunks + 1, np.uint64)
    total = _lib.compress_chunked_host(_CODEC_ID[codec], esz, _np_ptr(data), data.size, chunk_len, ndims,
                                       _np_ptr(comp), cap, _np_ptr(offsets))
    _lib.check(total)
    return comp[:total].copy(), offsets


def decompress_chunked(codec, comp, offsets, esz, ndims, chunk_len):
    comp = np.ascontiguousarray(comp, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    nchunks = len(offsets) - 1
    out = np.empty(nchunks * chunk_len, _NP[esz])
    n = _lib.decompress_chunked_host(_CODEC_ID[codec], esz, _np_ptr(comp), _np_ptr(offsets), nchunks, chunk_len, ndims,
                                     _np_ptr(out))
    _lib.check(n)
    return out[:n]

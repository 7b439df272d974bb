"""ctypes binding of libsprintz_mi355x.so (C-ABI: include/sprintz_mi355x.h).

The library is the product; this module only loads it and declares
signatures.  There is no Python or CPU fallback: if the shared object is
missing the import fails loudly, and every entry point returns
SPRINTZ_E_NO_DEVICE when no HIP device is usable.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPRINTZ_MI355X_LIB") or os.path.join(_HERE, "libsprintz_mi355x.so")   # env override: A/B builds

E_INVALID, E_NO_DEVICE, E_HIP, E_UNSUPPORTED, E_CORRUPT = -1, -2, -3, -4, -5
CODEC_DELTA, CODEC_XFF, CODEC_DELTA_NORLE, CODEC_BITPACK_NORLE, CODEC_XFF_NORLE = 0, 1, 2, 3, 4
READ_SLACK = 16
MAX_NDIMS = 65535


class SprintzError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libsprintz_mi355x error {code}: {msg}")
        self.code = code


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C sprintz_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # Share torch's HIP runtime when torch is in the process: both libraries
    # resolve libamdhip64.so.7 by SONAME, so importing torch first makes device
    # pointers and streams interchangeable.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for pure ctypes users
        pass
    return C.CDLL(LIB_PATH)


lib = _load()

_vp, _i, _u16, _u32, _u64, _i64, _sz = C.c_void_p, C.c_int, C.c_uint16, C.c_uint32, C.c_uint64, C.c_int64, C.c_size_t


def _sig(name, restype, *argtypes):
    f = getattr(lib, name)
    f.restype = restype
    f.argtypes = list(argtypes)
    return f


abi_version = _sig("sprintz_mi355x_abi_version", _i)
ABI_REQUIRED = 7
if abi_version() < ABI_REQUIRED:      # a stale build would otherwise die below with an AttributeError on the first new symbol
    raise ImportError(f"{LIB_PATH} has ABI version {abi_version()}, this binding needs >= {ABI_REQUIRED}: rebuild it "
                      "(`make -C sprintz_amd/csrc`)")
_last_error = _sig("sprintz_mi355x_last_error", C.c_char_p)
OPT_NO_FAST, OPT_CHUNKS_PER_GROUP, OPT_DENSE_MODE, OPT_HUF0_BIG_BATCH, OPT_SPLIT_LANES, OPT_ENC_PAIR, OPT_HOST_WAIT, OPT_LAT_CHUNKS, OPT_HOST_STREAMS, OPT_REF_DECODER_QUIRK, OPT_HUF0_SYNC_CHUNKS, OPT_BLK_CHUNKS, OPT_BLK_KERNELS = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12
set_option = _sig("sprintz_mi355x_set_option", _i, _i, _i)

# (1) drop-in single-call API, host pointers
compress = {
    ("delta", 1): _sig("sprintz_mi355x_compress_delta_8b", _i64, _vp, _u32, _vp, _u16, _i),
    ("xff", 1): _sig("sprintz_mi355x_compress_xff_8b", _i64, _vp, _u32, _vp, _u16, _i),
    ("delta", 2): _sig("sprintz_mi355x_compress_delta_16b", _i64, _vp, _u32, _vp, _u16, _i),
    ("xff", 2): _sig("sprintz_mi355x_compress_xff_16b", _i64, _vp, _u32, _vp, _u16, _i),
}
decompress = {
    ("delta", 1): _sig("sprintz_mi355x_decompress_delta_8b", _i64, _vp, _vp),
    ("xff", 1): _sig("sprintz_mi355x_decompress_xff_8b", _i64, _vp, _vp),
    ("delta", 2): _sig("sprintz_mi355x_decompress_delta_16b", _i64, _vp, _vp),
    ("xff", 2): _sig("sprintz_mi355x_decompress_xff_16b", _i64, _vp, _vp),
}
decompress_noheader = _sig("sprintz_mi355x_decompress_noheader", _i64, _i, _i, _vp, _vp, _u16, _u32, _u16)
# the layer below sprintz.h: payload layout chosen by the caller (sprintz_delta.h:49-91, sprintz_xff.h:43-85)
LAYOUT_AUTO, LAYOUT_GENERAL, LAYOUT_LOWDIM = 0, 1, 2
compress_layout = _sig("sprintz_mi355x_compress_layout", _i64, _i, _i, _vp, _u32, _vp, _u16, _i, _i)
decompress_layout = _sig("sprintz_mi355x_decompress_layout", _i64, _i, _i, _vp, _vp, _i)

# (2) batched device API
compress_bound = _sig("sprintz_mi355x_compress_bound", _sz, _i, _u32, _u16)
num_chunks = _sig("sprintz_mi355x_num_chunks", _u64, _u64, _u32)
compress_batch = _sig("sprintz_mi355x_compress_batch", _i, _i, _i, _vp, _u64, _u32, _u16, _vp, _sz, _vp, _vp, _vp)
compact_tmp_bytes = _sig("sprintz_mi355x_compact_tmp_bytes", _sz, _u64)
compress_dense_tmp_bytes = _sig("sprintz_mi355x_compress_dense_tmp_bytes", _sz, _u64)
compress_batch_dense = _sig("sprintz_mi355x_compress_batch_dense", _i, _i, _i, _vp, _u64, _u32, _u16, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp)
compact = _sig("sprintz_mi355x_compact", _i, _vp, _sz, _vp, _u64, _u32, _vp, _vp, _vp, _vp)
decompress_batch = _sig("sprintz_mi355x_decompress_batch", _i, _i, _i, _vp, _vp, _u64, _u32, _u16, _vp, _vp, _vp)

# (3) optional Huffman stage
huf_tmp_bytes = _sig("sprintz_mi355x_huf_tmp_bytes", _sz, _u64)
huf_bound = _sig("sprintz_mi355x_huf_bound", _sz, _u64, _u64)
huf_compress_batch = _sig("sprintz_mi355x_huf_compress_batch", _i, _vp, _vp, _vp, _u64, _vp, _vp, _vp, _vp, _vp)
huf0_tmp_bytes = _sig("sprintz_mi355x_huf0_tmp_bytes", _sz, _u64)
huf0_bound = _sig("sprintz_mi355x_huf0_bound", _sz, _u64, _u64)
huf0_compress_batch = _sig("sprintz_mi355x_huf0_compress_batch", _i, _vp, _vp, _vp, _u64, _vp, _vp, _vp, _vp)
huf0_decompress_batch = _sig("sprintz_mi355x_huf0_decompress_batch", _i, _vp, _vp, _u64, _vp, _vp, _vp, _vp)
huf0_decode_tmp_bytes = _sig("sprintz_mi355x_huf0_decode_tmp_bytes", _sz, _u64)
huf0_decompress_batch_ws = _sig("sprintz_mi355x_huf0_decompress_batch_ws", _i, _vp, _vp, _u64, _vp, _vp, _vp, _vp, _vp)
huf0_decompress_batch_hint = _sig("sprintz_mi355x_huf0_decompress_batch_hint", _i, _vp, _vp, _u64, _vp, _vp, _vp, _vp, _u32, _vp)
huf_decompress_batch = _sig("sprintz_mi355x_huf_decompress_batch", _i, _vp, _vp, _vp, _u64, _u32, _vp, _u64, _vp, _vp, _vp, _vp, _vp)

# (4) query on compressed data (query.hpp:23-29; sprintz_delta.h:95-98; sprintz_xff.h:90-93)
QUERY_NOOP, QUERY_MAX, QUERY_SUM = 0, 1, 2
QUERY_GENERAL_LAYOUT = 1
query_batch = _sig("sprintz_mi355x_query_batch", _i, _i, _i, _vp, _vp, _u64, _u32, _u16, _i, _i, _u32, _vp, _vp, _vp, _vp)
query_reduce = _sig("sprintz_mi355x_query_reduce", _i, _i, _vp, _u64, _u16, _vp, _vp)
query = {
    ("delta", 1): _sig("sprintz_mi355x_query_delta_8b", _i64, _vp, _vp, _i, _i, _u32, _vp),
    ("xff", 1): _sig("sprintz_mi355x_query_xff_8b", _i64, _vp, _vp, _i, _i, _u32, _vp),
    ("delta", 2): _sig("sprintz_mi355x_query_delta_16b", _i64, _vp, _vp, _i, _i, _u32, _vp),
    ("xff", 2): _sig("sprintz_mi355x_query_xff_16b", _i64, _vp, _vp, _i, _i, _u32, _vp),
}

# (5) column-major matrices (BASELINE config 5)
compress_batch_colmajor = _sig("sprintz_mi355x_compress_batch_colmajor", _i, _i, _i, _vp, _u64, _u64, _u32, _u16, _vp, _sz, _vp, _vp, _vp)
compress_batch_colmajor_dense = _sig("sprintz_mi355x_compress_batch_colmajor_dense", _i, _i, _i, _vp, _u64, _u64, _u32, _u16, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp)
decompress_batch_colmajor = _sig("sprintz_mi355x_decompress_batch_colmajor", _i, _i, _i, _vp, _vp, _u64, _u32, _u16, _u64, _vp, _vp, _vp)

# (6) stand-alone transforms (delta.h:17-68)
TRANSFORM_DELTA, TRANSFORM_DOUBLEDELTA, TRANSFORM_XFF = 0, 1, 2
transform_tmp_bytes = _sig("sprintz_mi355x_transform_tmp_bytes", _sz, _i, _i, _u64, _u16)
transform_encode_device = _sig("sprintz_mi355x_transform_encode_device", _i, _i, _i, _vp, _u64, _u16, _vp, _vp)
transform_decode_device = _sig("sprintz_mi355x_transform_decode_device", _i, _i, _i, _vp, _u64, _u16, _vp, _vp, _vp)
transform_encode = _sig("sprintz_mi355x_transform_encode", _i64, _i, _i, _vp, _u32, _vp, _u16, _i)
transform_decode = _sig("sprintz_mi355x_transform_decode", _i64, _i, _i, _vp, _vp, _u32, _u16)
_transform_last_error = _sig("sprintz_mi355x_transform_last_error", C.c_char_p)

# (7) non-RLE codecs, single call (sprintz_delta.h:26-76)
compress_norle = _sig("sprintz_mi355x_compress_norle", _i64, _i, _i, _vp, _u32, _vp, _u16)
decompress_norle = _sig("sprintz_mi355x_decompress_norle", _i64, _i, _i, _vp, _vp)

# (8) the reference's 2020 "online" u16 coders (online.hpp:395-445)
ONLINE_DYNDELTA, ONLINE_DYNDELTA_ALT, ONLINE_ZIGZAG, ONLINE_PACK, ONLINE_PACK_ZIGZAG = 0, 1, 2, 3, 4
online_bound = _sig("sprintz_mi355x_online_bound", _sz, _i, _u32)
online_tmp_bytes = _sig("sprintz_mi355x_online_tmp_bytes", _sz, _i, _u32)
online_pack_device = _sig("sprintz_mi355x_online_pack_device", _i, _i, _vp, _u32, _vp, _vp, _vp, _vp)
online_unpack_device = _sig("sprintz_mi355x_online_unpack_device", _i, _i, _vp, _u32, _vp, _vp, _vp, _vp)
online_pack = _sig("sprintz_mi355x_online_pack", _i64, _i, _vp, _u32, _vp)
online_unpack = _sig("sprintz_mi355x_online_unpack", _i64, _i, _vp, _vp)

# multi-GPU: the all-gather of per-rank byte counts over RCCL
COMM_ID_BYTES = 128
comm_unique_id = _sig("sprintz_mi355x_comm_unique_id", _i, _vp)
comm_init = _sig("sprintz_mi355x_comm_init", _i, _vp, _i, _i, C.POINTER(C.c_void_p))
gather_layout = _sig("sprintz_mi355x_gather_layout", _i, _vp, _vp, _vp, _vp)
layout_bases = _sig("sprintz_mi355x_layout_bases", _i, _vp, _i, _vp, _vp)
comm_destroy = _sig("sprintz_mi355x_comm_destroy", _i, _vp)

# host convenience
compress_chunked_host = _sig("sprintz_mi355x_compress_chunked_host", _i64, _i, _i, _vp, _u64, _u32, _u16, _vp, _sz, _vp)
decompress_chunked_host = _sig("sprintz_mi355x_decompress_chunked_host", _i64, _i, _i, _vp, _vp, _u64, _u32, _u16, _vp)

EXPORTED_SYMBOLS = [
    "sprintz_mi355x_abi_version", "sprintz_mi355x_last_error", "sprintz_mi355x_set_option",
    "sprintz_mi355x_compress_delta_8b", "sprintz_mi355x_compress_xff_8b",
    "sprintz_mi355x_compress_delta_16b", "sprintz_mi355x_compress_xff_16b",
    "sprintz_mi355x_decompress_delta_8b", "sprintz_mi355x_decompress_xff_8b",
    "sprintz_mi355x_decompress_delta_16b", "sprintz_mi355x_decompress_xff_16b",
    "sprintz_mi355x_decompress_noheader", "sprintz_mi355x_compress_layout", "sprintz_mi355x_decompress_layout",
    "sprintz_mi355x_compress_bound", "sprintz_mi355x_num_chunks",
    "sprintz_mi355x_compress_batch", "sprintz_mi355x_compact_tmp_bytes", "sprintz_mi355x_compact",
    "sprintz_mi355x_compress_dense_tmp_bytes", "sprintz_mi355x_compress_batch_dense",
    "sprintz_mi355x_decompress_batch",
    "sprintz_mi355x_compress_chunked_host", "sprintz_mi355x_decompress_chunked_host",
    "sprintz_mi355x_online_bound", "sprintz_mi355x_online_tmp_bytes", "sprintz_mi355x_online_pack_device",
    "sprintz_mi355x_online_unpack_device", "sprintz_mi355x_online_pack", "sprintz_mi355x_online_unpack",
    "sprintz_mi355x_comm_unique_id", "sprintz_mi355x_comm_init", "sprintz_mi355x_gather_layout",
    "sprintz_mi355x_layout_bases", "sprintz_mi355x_comm_destroy",
    "sprintz_mi355x_huf_tmp_bytes", "sprintz_mi355x_huf_bound",
    "sprintz_mi355x_huf_compress_batch", "sprintz_mi355x_huf_decompress_batch", "sprintz_mi355x_huf0_decompress_batch",
    "sprintz_mi355x_huf0_decode_tmp_bytes", "sprintz_mi355x_huf0_decompress_batch_ws", "sprintz_mi355x_huf0_decompress_batch_hint",
    "sprintz_mi355x_huf0_tmp_bytes", "sprintz_mi355x_huf0_bound", "sprintz_mi355x_huf0_compress_batch",
    "sprintz_mi355x_query_batch", "sprintz_mi355x_query_reduce",
    "sprintz_mi355x_query_delta_8b", "sprintz_mi355x_query_xff_8b",
    "sprintz_mi355x_query_delta_16b", "sprintz_mi355x_query_xff_16b",
    "sprintz_mi355x_compress_batch_colmajor", "sprintz_mi355x_compress_batch_colmajor_dense", "sprintz_mi355x_decompress_batch_colmajor",
    "sprintz_mi355x_transform_tmp_bytes", "sprintz_mi355x_transform_encode_device", "sprintz_mi355x_transform_decode_device",
    "sprintz_mi355x_transform_encode", "sprintz_mi355x_transform_decode", "sprintz_mi355x_transform_last_error",
    "sprintz_mi355x_compress_norle", "sprintz_mi355x_decompress_norle",
]


_CODE_NAMES = {E_INVALID: "invalid argument", E_NO_DEVICE: "no usable HIP device (there is no CPU fallback)",
               E_HIP: "a HIP call failed", E_UNSUPPORTED: "unsupported argument", E_CORRUPT: "corrupt stream"}


def last_error():
    """message of this thread's last failing call (every failing return of the library sets it)"""
    return _last_error().decode("utf-8", "replace")


def check(rc):
    """raise on negative return codes of the batched API"""
    if rc < 0:
        raise SprintzError(rc, last_error() or _CODE_NAMES.get(rc, "error"))
    return rc

"""sprintz_amd -- MI355X-native Sprintz codec hot path.

The product is ``libsprintz_mi355x.so`` (C-ABI: include/sprintz_mi355x.h, HIP
kernels in sprintz_amd/csrc).  This package is the host-side mirror of the
reference's interface (cpp/Compress/sprintz.h) on top of it.
"""
from . import _lib
from ._lib import SprintzError, abi_version, last_error
from .codec import (ChunkedCodec, CompressedBatch, HufBatch, compress_chunked, decompress_chunked, decompress_noheader,
                    huf_compress, huf_decompress, huf0_compress, huf0_decompress, QueryParams, QueryTypes,
                    query_rowmajor_delta_rle_8b, query_rowmajor_delta_rle_16b, query_rowmajor_xff_rle_8b,
                    query_rowmajor_xff_rle_16b,
                    encode_delta_rowmajor_8b, encode_delta_rowmajor_16b, encode_doubledelta_rowmajor_8b,
                    encode_doubledelta_rowmajor_16b, decode_delta_rowmajor_8b, decode_delta_rowmajor_16b,
                    decode_doubledelta_rowmajor_8b, decode_doubledelta_rowmajor_16b, transform_device,
                    encode_xff_rowmajor_8b, encode_xff_rowmajor_16b, decode_xff_rowmajor_8b, decode_xff_rowmajor_16b,
                    compress_rowmajor_8b, compress_rowmajor_16b, compress_rowmajor_delta_8b, compress_rowmajor_delta_16b,
                    decompress_rowmajor_8b, decompress_rowmajor_16b, decompress_rowmajor_delta_8b, decompress_rowmajor_delta_16b,
                    compress8b_rowmajor_xff, decompress8b_rowmajor_xff,
                    sprintz_compress_delta_8b, sprintz_compress_delta_16b, sprintz_compress_xff_8b,
                    sprintz_compress_xff_16b, sprintz_decompress_delta_8b, sprintz_decompress_delta_16b,
                    sprintz_decompress_xff_8b, sprintz_decompress_xff_16b)

__all__ = [
    "SprintzError", "abi_version", "last_error", "ChunkedCodec", "CompressedBatch", "HufBatch", "huf_compress", "huf_decompress", "huf0_compress", "huf0_decompress",
    "compress_chunked", "decompress_chunked", "decompress_noheader", "QueryParams", "QueryTypes",
    "encode_delta_rowmajor_8b", "encode_delta_rowmajor_16b", "encode_doubledelta_rowmajor_8b", "encode_doubledelta_rowmajor_16b",
    "decode_delta_rowmajor_8b", "decode_delta_rowmajor_16b", "decode_doubledelta_rowmajor_8b", "decode_doubledelta_rowmajor_16b",
    "transform_device", "encode_xff_rowmajor_8b", "encode_xff_rowmajor_16b", "decode_xff_rowmajor_8b", "decode_xff_rowmajor_16b",
    "compress_rowmajor_8b", "compress_rowmajor_16b", "compress_rowmajor_delta_8b", "compress_rowmajor_delta_16b",
    "decompress_rowmajor_8b", "decompress_rowmajor_16b", "decompress_rowmajor_delta_8b", "decompress_rowmajor_delta_16b",
    "compress8b_rowmajor_xff", "decompress8b_rowmajor_xff",
    "query_rowmajor_delta_rle_8b", "query_rowmajor_delta_rle_16b", "query_rowmajor_xff_rle_8b", "query_rowmajor_xff_rle_16b",
    "sprintz_compress_delta_8b", "sprintz_compress_delta_16b", "sprintz_compress_xff_8b", "sprintz_compress_xff_16b",
    "sprintz_decompress_delta_8b", "sprintz_decompress_delta_16b", "sprintz_decompress_xff_8b",
    "sprintz_decompress_xff_16b",
]

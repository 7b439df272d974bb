"""Link-level drop-in, executed on the GPU: a C++ program written against the REFERENCE's prototypes
(declared inline below with the reference's signatures -- cpp/Compress/sprintz.h:16-32 and the
headers it sits on; nothing from include/ is seen by the compiler) is linked against
libsprintz_mi355x.so by the reference's mangled names, run on the device, and its output compared
byte for byte with the golden vectors minted from the compiled reference.  Also pushes the
reference test-suite's long stream (test/compress_testing.hpp:445-461, 1024*1024+7 elements)
through the single-call symbols."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from harness import CODECS, gen_fuzz, gen_sparse

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("decode_path")]

# The caller: a job file in, a result file out.  Prototypes exactly as the reference declares them.
CALLER = r"""
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
// ---- cpp/Compress/sprintz.h:16-32 (C++ linkage, default argument)
int64_t sprintz_compress_delta_8b(const uint8_t* src, uint32_t len, int8_t* dest, uint16_t ndims, bool write_size=true);
int64_t sprintz_decompress_delta_8b(const int8_t* src, uint8_t* dest);
int64_t sprintz_compress_xff_8b(const uint8_t* src, uint32_t len, int8_t* dest, uint16_t ndims, bool write_size=true);
int64_t sprintz_decompress_xff_8b(const int8_t* src, uint8_t* dest);
int64_t sprintz_compress_delta_16b(const uint16_t* src, uint32_t len, int16_t* dest, uint16_t ndims, bool write_size=true);
int64_t sprintz_decompress_delta_16b(const int16_t* src, uint16_t* dest);
int64_t sprintz_compress_xff_16b(const uint16_t* src, uint32_t len, int16_t* dest, uint16_t ndims, bool write_size=true);
int64_t sprintz_decompress_xff_16b(const int16_t* src, uint16_t* dest);
// ---- cpp/Compress/sprintz_delta.h:49-56, sprintz_xff.h:53-60,66-73 (the layer below)
int64_t compress_rowmajor_delta_rle_8b(const uint8_t* src, uint32_t len, int8_t* dest, uint16_t ndims, bool write_size=true);
int64_t decompress_rowmajor_delta_rle_8b(const int8_t* src, uint8_t* dest);
int64_t compress_rowmajor_xff_rle_16b(const uint16_t* src, uint32_t len, int16_t* dest, uint16_t ndims, bool write_size=true);
int64_t decompress_rowmajor_xff_rle_16b(const int16_t* src, uint16_t* dest);
int64_t compress_rowmajor_xff_rle_lowdim_8b(const uint8_t* src, uint32_t len, int8_t* dest, uint16_t ndims, bool write_size=true);
int64_t decompress_rowmajor_xff_rle_lowdim_8b(const int8_t* src, uint8_t* dest);
// ---- cpp/Compress/query.hpp:23-29, sprintz_xff.h:90-93
namespace QueryTypes { enum Operation { NOOP = 0, REDUCE_MAX, REDUCE_SUM }; }
typedef struct QueryParams { QueryTypes::Operation op; bool materialize; } QueryParams;
int64_t query_rowmajor_xff_rle_16b(const int16_t* src, uint16_t* dest, const QueryParams& qparams);
// ---- cpp/Compress/delta.h:52-58
uint32_t encode_delta_rowmajor_16b(const uint16_t* src, uint32_t len, int16_t* dest, uint16_t ndims, bool write_size=true);
uint32_t decode_delta_rowmajor_16b(const int16_t* src, uint16_t* dest);

// ---- cpp/Compress/online.hpp:15,412-462
typedef uint32_t len_t;
len_t dynamic_delta_pack_u16(const uint16_t* data_in, size_t length, int16_t* data_out);
len_t dynamic_delta_unpack_u16(const int16_t* data_in, uint16_t* data_out);
len_t sprintzpack_pack_u16_zigzag(const uint16_t* data_in, size_t length, int16_t* data_out);
len_t sprintzpack_unpack_u16_zigzag(const int16_t* data_in, uint16_t* data_out);

struct Job { uint8_t kind, esz; uint16_t ndims; uint32_t len; };   // kind: 0 delta, 1 xff (sprintz.h); 2.. see below

int main(int argc, char** argv)
{
    if (argc != 3) return 2;
    FILE* fi = fopen(argv[1], "rb");
    FILE* fo = fopen(argv[2], "wb");
    if (!fi || !fo) return 3;
    Job j;
    while (fread(&j, sizeof j, 1, fi) == 1) {
        std::vector<uint8_t> in((size_t)j.len * j.esz + 64), out((size_t)j.len * j.esz + 64 * j.esz + 64);
        std::vector<uint8_t> comp(((size_t)j.len * 3 / 2 + 64) * j.esz + 64);      // test/compress_testing.hpp:145-147
        if (j.len && fread(in.data(), (size_t)j.len * j.esz, 1, fi) != 1) return 4;
        int64_t cret = -99, dret = -99;
        const uint8_t* s8 = in.data(); const uint16_t* s16 = (const uint16_t*)in.data();
        int8_t* c8 = (int8_t*)comp.data(); int16_t* c16 = (int16_t*)comp.data();
        uint8_t* o8 = out.data(); uint16_t* o16 = (uint16_t*)out.data();
        switch (j.kind) {
        case 0: if (j.esz == 1) { cret = sprintz_compress_delta_8b(s8, j.len, c8, j.ndims); dret = sprintz_decompress_delta_8b(c8, o8); }
                else            { cret = sprintz_compress_delta_16b(s16, j.len, c16, j.ndims); dret = sprintz_decompress_delta_16b(c16, o16); } break;
        case 1: if (j.esz == 1) { cret = sprintz_compress_xff_8b(s8, j.len, c8, j.ndims); dret = sprintz_decompress_xff_8b(c8, o8); }
                else            { cret = sprintz_compress_xff_16b(s16, j.len, c16, j.ndims); dret = sprintz_decompress_xff_16b(c16, o16); } break;
        case 2: cret = compress_rowmajor_delta_rle_8b(s8, j.len, c8, j.ndims); dret = decompress_rowmajor_delta_rle_8b(c8, o8); break;
        case 3: cret = compress_rowmajor_xff_rle_16b(s16, j.len, c16, j.ndims); dret = decompress_rowmajor_xff_rle_16b(c16, o16); break;
        case 4: cret = compress_rowmajor_xff_rle_lowdim_8b(s8, j.len, c8, j.ndims); dret = decompress_rowmajor_xff_rle_lowdim_8b(c8, o8); break;
        case 5: {   // query with materialize == true must reproduce the data (test/test_query.cpp:59-120)
            cret = compress_rowmajor_xff_rle_16b(s16, j.len, c16, j.ndims);
            QueryParams qp; qp.op = QueryTypes::REDUCE_MAX; qp.materialize = true;
            dret = query_rowmajor_xff_rle_16b(c16, o16, qp); break; }
        case 6: cret = encode_delta_rowmajor_16b(s16, j.len, c16, j.ndims); dret = decode_delta_rowmajor_16b(c16, o16); break;
        case 7: cret = sprintz_compress_xff_16b(s16, j.len, c16, j.ndims, false); dret = 0; break;   // headerless
        case 8: cret = dynamic_delta_pack_u16(s16, j.len, c16); dret = dynamic_delta_unpack_u16(c16, o16); break;
        case 9: cret = sprintzpack_pack_u16_zigzag(s16, j.len, c16); dret = sprintzpack_unpack_u16_zigzag(c16, o16); break;
        default: return 5;
        }
        // stream bytes: the return value is in elements (floor'ed), so report the exact byte count by
        // scanning back from the capacity for the last byte written -- the buffer starts zeroed
        size_t nb = comp.size();
        while (nb > 0 && comp[nb - 1] == 0) nb--;
        uint64_t nb64 = nb;
        fwrite(&cret, 8, 1, fo); fwrite(&dret, 8, 1, fo); fwrite(&nb64, 8, 1, fo);
        fwrite(comp.data(), 1, nb, fo);
        const uint64_t no = dret > 0 ? (uint64_t)dret * j.esz : 0;
        fwrite(out.data(), 1, no, fo);
    }
    fclose(fi); fclose(fo);
    return 0;
}
"""


@pytest.fixture(scope="module")
def caller(tmp_path_factory):
    if not shutil.which("g++"):
        pytest.skip("no g++")
    d = tmp_path_factory.mktemp("dropin")
    src = d / "caller.cpp"
    src.write_text(CALLER)
    exe = d / "caller"
    libdir = os.path.join(ROOT, "sprintz_amd")
    subprocess.check_call(["g++", "-std=c++14", "-O1", str(src), "-o", str(exe), "-L", libdir, "-lsprintz_mi355x",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"])
    return exe, d


def run_jobs(caller, jobs):
    """jobs: list of (kind, esz, ndims, data ndarray) -> list of (cret, dret, stream bytes, decoded array)"""
    exe, d = caller
    fin, fout = d / "jobs.bin", d / "res.bin"
    with open(fin, "wb") as f:
        for kind, esz, ndims, data in jobs:
            f.write(struct.pack("<BBHI", kind, esz, ndims, data.size))
            f.write(np.ascontiguousarray(data).tobytes())
    subprocess.check_call([str(exe), str(fin), str(fout)], timeout=600)
    raw = open(fout, "rb").read()
    res, p = [], 0
    for kind, esz, ndims, data in jobs:
        cret, dret, nb = struct.unpack_from("<qqQ", raw, p)
        p += 24
        stream = np.frombuffer(raw, np.uint8, nb, p)
        p += nb
        no = dret * esz if dret > 0 else 0
        dec = np.frombuffer(raw, np.uint8 if esz == 1 else np.uint16, no // esz, p)
        p += no
        res.append((cret, dret, stream, dec))
    assert p == len(raw)
    return res


def trimmed(a):
    """golden stream with trailing zero bytes removed (the caller cannot see them either)"""
    n = a.size
    while n > 0 and a[n - 1] == 0:
        n -= 1
    return a[:n]


def test_reference_prototypes_resolve_by_mangled_name():
    """the library exports the reference's C++ names next to the C ones"""
    out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(ROOT, "sprintz_amd", "libsprintz_mi355x.so")]).decode()
    for sym in ("_Z25sprintz_compress_delta_8bPKhjPatb", "_Z27sprintz_decompress_delta_8bPKaPh",
                "_Z24sprintz_compress_xff_16bPKtjPstb", "_Z26sprintz_decompress_xff_16bPKsPt",
                "_Z26query_rowmajor_xff_rle_16bPKsPtRK11QueryParams"):
        assert sym in out, sym


def test_golden_vectors_through_the_reference_abi(caller, golden):
    manifest, arrays = golden
    jobs = [(CODECS[m["codec"]], m["esz"], m["ndims"], arrays[f"in_{m['idx']}"]) for m in manifest]
    res = run_jobs(caller, jobs)
    for m, (cret, dret, stream, dec) in zip(manifest, res):
        want = arrays[f"out_{m['idx']}"]
        data = arrays[f"in_{m['idx']}"]
        assert cret == m["ret"], m
        assert np.array_equal(stream, trimmed(want)), m
        assert dret == data.size and np.array_equal(dec, data), m


def test_long_streams_of_the_reference_suite(caller, oracle):
    """test_codec's 1024*1024+7-element fuzz and sparse inputs (compress_testing.hpp:445-461)"""
    rng = np.random.default_rng(7)
    n = 1024 * 1024 + 7
    jobs = []
    for esz in (1, 2):
        for ndims in (1, 8, 80) if esz == 1 else (2, 8):
            jobs.append((1, esz, ndims, gen_fuzz(rng, n, esz, 1)))
            jobs.append((0, esz, ndims, gen_sparse(rng, n, esz, 0.05)))
    res = run_jobs(caller, jobs)
    for (kind, esz, ndims, data), (cret, dret, stream, dec) in zip(jobs, res):
        want, wret = oracle.compress("xff" if kind == 1 else "delta", data, ndims)
        assert cret == wret, (kind, esz, ndims)
        assert np.array_equal(stream, trimmed(want)), (kind, esz, ndims)
        assert dret == n and np.array_equal(dec, data), (kind, esz, ndims)


def test_layer_below_sprintz_h(caller, oracle):
    """compress_rowmajor_*_rle[_lowdim]_*: layout chosen by name (general layout for small ndims too),
    query(materialize) reproduces the data, the delta transform round-trips, write_size=false"""
    rng = np.random.default_rng(11)
    jobs = []
    for nd in (1, 2, 3, 4, 7):
        jobs.append((2, 1, nd, gen_fuzz(rng, 16 * nd * 9 + 5, 1, 2)))         # general layout, 8-bit delta
    for nd in (1, 2, 5, 8):
        jobs.append((3, 2, nd, gen_fuzz(rng, 16 * nd * 9 + 3, 2, 3)))         # general layout, 16-bit xff
        jobs.append((5, 2, nd, gen_fuzz(rng, 16 * nd * 9 + 3, 2, 3)))         # + query, materialized
    for nd in (1, 3, 4):
        jobs.append((4, 1, nd, gen_fuzz(rng, 16 * nd * 9 + 1, 1, 2)))         # low-dim layout by name
    jobs.append((4, 1, 5, gen_fuzz(rng, 1000, 1, 2)))                          # invalid for the low-dim codec: -1
    jobs.append((6, 2, 4, gen_fuzz(rng, 4 * 300, 2, 2)))
    jobs.append((7, 2, 8, gen_fuzz(rng, 8 * 64, 2, 3)))
    jobs.append((8, 2, 1, gen_fuzz(rng, 5003, 2, 7)))                          # online.hpp names
    jobs.append((9, 2, 1, gen_fuzz(rng, 4099, 2, 9)))
    res = run_jobs(caller, jobs)
    for (kind, esz, nd, data), (cret, dret, stream, dec) in zip(jobs, res):
        if kind == 4 and nd == 5:
            assert cret == -1 and dret == -1
            continue
        if kind in (2, 3, 5):
            want, wret = oracle.compress_rowmajor("delta" if kind == 2 else "xff", data, nd)
            assert cret == wret and np.array_equal(stream, trimmed(want)), (kind, nd)
        if kind == 4:
            want, wret = oracle.compress("xff", data, nd)
            assert cret == wret and np.array_equal(stream, trimmed(want)), (kind, nd)
        if kind == 7:
            want, wret = oracle.compress("xff", data, nd, write_size=False)
            assert cret == wret and np.array_equal(stream, trimmed(want))
            continue
        if kind in (8, 9):
            from test_online_cpu import oracle_pack
            import ctypes as C
            from harness import ORACLE_SO
            lib = C.CDLL(ORACLE_SO)
            lib.online_oracle_pack.restype = C.c_int64
            lib.online_oracle_pack.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_size_t)]
            lib.online_oracle_bound.restype = C.c_size_t
            lib.online_oracle_bound.argtypes = [C.c_int, C.c_uint32]
            want, wret, _ = oracle_pack(lib, 0 if kind == 8 else 4, data)
            assert cret == wret and np.array_equal(stream, trimmed(want)), kind
        assert dret == data.size and np.array_equal(dec, data), (kind, nd)

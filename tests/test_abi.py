"""CPU tests of the boundary: the C-ABI library loads, exports every symbol
include/sprintz_mi355x.h declares, and fails loudly without a GPU (no compute)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    p = os.path.join(ROOT, "sprintz_amd", "libsprintz_mi355x.so")
    if not os.path.exists(p):
        import __graft_entry__
        __graft_entry__.build()
    return p


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "sprintz_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sprintz_mi355x_\w+)\s*\(", text)))


def test_header_declares_the_reference_surface():
    syms = declared_symbols()
    for codec in ("delta", "xff"):
        for w in ("8b", "16b"):
            assert f"sprintz_mi355x_compress_{codec}_{w}" in syms
            assert f"sprintz_mi355x_decompress_{codec}_{w}" in syms
    for s in ("sprintz_mi355x_compress_batch", "sprintz_mi355x_decompress_batch", "sprintz_mi355x_compact"):
        assert s in syms


def test_library_exports_every_declared_symbol(lib_path):
    lib = C.CDLL(lib_path)
    for s in declared_symbols():
        assert hasattr(lib, s), f"{s} declared in include/sprintz_mi355x.h but not exported"


def test_python_binding_matches_header(lib_path):
    from sprintz_amd import _lib
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared_symbols()
    assert _lib.abi_version() == 7


def test_no_cpu_fallback(lib_path):
    """Without a GPU every entry point must fail loudly, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import sprintz_amd
    src = np.arange(4096, dtype=np.uint16)
    dest = np.zeros(4096 * 3 // 2 + 64, np.int16)
    rc = sprintz_amd.sprintz_compress_xff_16b(src, src.size, dest, 8)
    assert rc == sprintz_amd._lib.E_NO_DEVICE
    assert "no CPU fallback" in sprintz_amd.last_error()
    assert not dest.any()
    with pytest.raises(sprintz_amd.SprintzError):
        sprintz_amd.ChunkedCodec("xff", 2, 8, 5120)


def test_ndims_zero_matches_reference(lib_path):
    """ndims == 0 -> -1 before touching the device (sprintz.cpp:36)"""
    import sprintz_amd
    dest = np.zeros(400, np.int8)
    assert sprintz_amd.sprintz_compress_delta_8b(np.zeros(100, np.uint8), 100, dest, 0) == -1


def test_compress_bound_dominates_oracle_sizes(lib_path, oracle):
    from sprintz_amd import _lib
    from harness import gen_fuzz
    rng = np.random.default_rng(0)
    for esz in (1, 2):
        for D in (1, 3, 5, 8, 17, 80, 129):
            for n in (1, 127, 128, 16 * D, 16 * D * 7 + 3, 5120):
                d = gen_fuzz(rng, n, esz, 0)
                for codec in ("delta", "xff"):
                    s, _ = oracle.compress(codec, d, D)
                    assert s.size + 16 <= _lib.compress_bound(esz, n, D)


def test_dropin_header_compiles_and_links(lib_path, tmp_path):
    """a caller written against the reference's sprintz.h builds against include/sprintz_dropin.hpp"""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    src = tmp_path / "caller.cpp"
    src.write_text(
        '#include "sprintz_dropin.hpp"\n'
        "#include <vector>\n"
        "int main() {\n"
        "  std::vector<uint16_t> x(4096, 7), y(4096 + 64);\n"
        "  std::vector<int16_t> c(4096 * 3 / 2 + 64);\n"
        "  int64_t n = sprintz_compress_xff_16b(x.data(), 4096, c.data(), 8);      // default write_size\n"
        "  int64_t m = sprintz_decompress_xff_16b(c.data(), y.data());\n"
        "  int64_t a = sprintz_compress_delta_8b((const uint8_t*)x.data(), 100, (int8_t*)c.data(), 3, false);\n"
        "  QueryParams qp; qp.op = QueryTypes::REDUCE_MAX; qp.materialize = false;   // query.hpp:23-29\n"
        "  uint64_t res[8];\n"
        "  int64_t q = query_rowmajor_xff_rle_16b(c.data(), y.data(), qp);           // sprintz_xff.h:92\n"
        "  int64_t r = query_rowmajor_delta_rle_8b((const int8_t*)c.data(), (uint8_t*)y.data(), qp, res);\n"
        "  uint32_t e = encode_doubledelta_rowmajor_16b(x.data(), 100, c.data(), 4);   // delta.h:63\n"
        "  uint32_t d = decode_delta_rowmajor_inplace_8b((uint8_t*)y.data(), 64, 2);     // delta.h:21\n"
        "  e += encode_xff_rowmajor_16b(x.data(), 100, c.data(), 4) + decode_xff_rowmajor_8b((const int8_t*)c.data(), (uint8_t*)y.data());   // predict.h:24,21\n"
        "  int64_t z = compress_rowmajor_delta_16b(x.data(), 4096, c.data(), 8) + decompress8b_rowmajor_xff((const int8_t*)c.data(), (uint8_t*)y.data());\n"
        "  (void)e; (void)d; (void)z;\n"
        "  return (n < 0 && m < 0 && a < 0 && q < 0 && r < 0) ? 0 : 1;   // without a GPU every call fails loudly\n"
        "}\n")
    exe = tmp_path / "caller"
    libdir = os.path.dirname(lib_path)
    cmd = ["g++", "-std=c++14", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
           "-L", libdir, "-lsprintz_mi355x", "-L/opt/rocm/lib", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    assert exe.exists()


def test_argument_checks_come_before_the_device_check(lib_path):
    """bad arguments are E_INVALID / E_UNSUPPORTED on any machine; good ones need a device"""
    import ctypes as C
    import torch
    from sprintz_amd import _lib
    E = _lib
    buf = (C.c_uint8 * 4096)()
    p = C.addressof(buf)
    p16 = (p + 15) & ~15
    # batched codec entry points
    assert _lib.compress_batch(9, 2, p16, 100, 50, 8, p16, 1024, p16, None, None) == E.E_INVALID          # codec
    assert _lib.compress_batch(0, 3, p16, 100, 50, 8, p16, 1024, p16, None, None) == E.E_INVALID          # elem_bytes
    assert _lib.compress_batch(0, 2, p16, 100, 50, 0, p16, 1024, p16, None, None) == E.E_INVALID          # ndims
    # (every uint16 ndims is taken since ABI 6: SPRINTZ_MI355X_MAX_NDIMS is 65535, what the stream header holds)
    assert _lib.compress_batch(4, 2, p16, 100, 50, 8, p16, 1024, p16, None, None) == E.E_UNSUPPORTED      # xff_norle is 8-bit only
    assert _lib.compress_batch(0, 2, p16, 100, 0, 8, p16, 1024, p16, None, None) == E.E_INVALID           # chunk_len
    assert _lib.compress_batch(0, 2, p16, 100, 50, 8, p16 + 4, 1024, p16, None, None) == E.E_INVALID      # slot alignment
    assert _lib.compress_batch(0, 2, p16, 100, 50, 8, p16, 16, p16, None, None) == E.E_INVALID            # slot_stride < bound
    assert _lib.decompress_batch(0, 2, None, p16, 1, 50, 8, p16, None, None) == E.E_INVALID               # null
    # query
    assert _lib.query_batch(1, 2, p16, p16, 1, 50, 8, 7, 0, 0, None, p16, None, None) == E.E_INVALID      # op
    assert _lib.query_batch(1, 2, p16, p16, 1, 50, 8, 1, 1, 0, None, p16, None, None) == E.E_INVALID      # materialize without out
    assert _lib.query_batch(1, 2, p16, p16, 1, 50, 8, 1, 0, 0, None, None, None, None) == E.E_INVALID     # op without partials
    assert _lib.query_batch(1, 2, p16, p16, 1, 50, 8, 1, 0, 8, None, p16, None, None) == E.E_INVALID      # unknown flag
    assert _lib.query_reduce(0, p16, 1, 8, p16, None) == E.E_INVALID
    # column-major
    assert _lib.compress_batch_colmajor(1, 2, p16, 100, 50, 10, 8, p16, 1024, p16, None, None) == E.E_INVALID   # col_stride < nrows
    assert _lib.decompress_batch_colmajor(1, 2, p16, p16, 4, 10, 8, 30, p16, None, None) == E.E_INVALID          # col_stride < nchunks*rows
    # transforms
    assert _lib.transform_encode_device(3, 2, p16, 10, 8, p16, None) == E.E_INVALID
    assert _lib.transform_encode_device(0, 4, p16, 10, 8, p16, None) == E.E_INVALID
    assert _lib.transform_decode_device(0, 2, p16, 10, 0, p16, p16, None) == E.E_INVALID
    assert _lib.transform_decode_device(0, 2, p16, 10, 8, p16, None, None) == E.E_INVALID
    assert _lib.transform_tmp_bytes(0, 2, 1 << 30, 8) > 0 and _lib.transform_tmp_bytes(0, 2, 100, 0) == 0
    # non-RLE single call
    assert _lib.compress_norle(1, 1, p16, 10, p16, 3) == E.E_INVALID
    # Huffman
    assert _lib.huf_compress_batch(None, p16, p16, 1, p16, p16, p16, p16, None) == E.E_INVALID
    assert _lib.huf_decompress_batch(p16, p16, p16, 1, 3, p16, 100, p16, p16, None, p16, None) == E.E_INVALID   # align not a power of two
    if not torch.cuda.is_available():
        assert _lib.compress_batch(0, 2, p16, 100, 50, 8, p16, 1024, p16, None, None) == E.E_NO_DEVICE
        assert _lib.transform_encode_device(0, 2, p16, 10, 8, p16, None) == E.E_NO_DEVICE
        assert _lib.query_batch(1, 2, p16, p16, 1, 50, 8, 1, 0, 0, None, p16, None, None) == E.E_NO_DEVICE


def test_unloadable_rccl_is_an_error_code_not_a_crash(lib_path):
    """comm.cpp resolves RCCL with dlopen; when that fails the entry points must return SPRINTZ_E_UNSUPPORTED with the
    loader's message (round 2 called dlerror() twice and fed the second call's NULL to std::string: a crash inside call_once)"""
    import subprocess
    import sys
    code = (
        "import ctypes as C, sys\n"
        f"lib = C.CDLL({lib_path!r})\n"
        "lib.sprintz_mi355x_last_error.restype = C.c_char_p\n"
        "buf = C.create_string_buffer(128)\n"
        "rc = lib.sprintz_mi355x_comm_unique_id(buf)\n"
        "comm = C.c_void_p()\n"
        "rc2 = lib.sprintz_mi355x_comm_init(buf, 0, 1, C.byref(comm))\n"
        "print(rc, rc2, lib.sprintz_mi355x_last_error().decode())\n")
    env = dict(os.environ, SPRINTZ_MI355X_RCCL_SONAME="/nonexistent/librccl-not-here.so")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-1500:]
    rc, rc2, msg = p.stdout.strip().split(" ", 2)
    assert int(rc) == -4 and int(rc2) == -4              # SPRINTZ_E_UNSUPPORTED
    assert "RCCL not loadable" in msg and "librccl-not-here" in msg


def test_options_are_validated_without_a_device():
    """sprintz_mi355x_set_option: every documented knob takes its documented values on a box without a GPU too, and an unknown
    option or a value out of range is SPRINTZ_E_INVALID (include/sprintz_mi355x.h: the tuning knobs)"""
    from sprintz_amd import _lib
    for opt, good, bad in [(_lib.OPT_NO_FAST, [0, 1], []), (_lib.OPT_CHUNKS_PER_GROUP, [1, 64], [0, 65]), (_lib.OPT_DENSE_MODE, [0, 1], [-1, 2]),
                           (_lib.OPT_HUF0_BIG_BATCH, [0, 16385], [-1]), (_lib.OPT_SPLIT_LANES, [0, 1], []), (_lib.OPT_ENC_PAIR, [0, 1, 1024], [-1])]:
        for v in good:
            assert _lib.set_option(opt, v) == 0, (opt, v)
        for v in bad:
            assert _lib.set_option(opt, v) == _lib.E_INVALID, (opt, v)
    assert _lib.set_option(99, 0) == _lib.E_INVALID
    for opt, v in [(_lib.OPT_NO_FAST, 0), (_lib.OPT_CHUNKS_PER_GROUP, 1), (_lib.OPT_DENSE_MODE, 1), (_lib.OPT_HUF0_BIG_BATCH, 16385),
                   (_lib.OPT_SPLIT_LANES, 1), (_lib.OPT_ENC_PAIR, 1)]:
        assert _lib.set_option(opt, v) == 0                  # back to the defaults

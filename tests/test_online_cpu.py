"""CPU tests of the "online" u16 coders' oracle (oracle/online_oracle.c; reference: cpp/Compress/online.hpp:395-445):
against golden containers minted from the compiled reference, and -- where oracle/_ref exists -- against the
compiled reference itself on a wider random set.  Bit-exact on every byte the reference writes (it leaves header
padding unwritten; those bytes are 0 in the fixtures and in our output)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from harness import ORACLE_SO, REF_SO, gen_fuzz, gen_walk

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def golden_online():
    gdir = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(gdir, "golden_online_v1.json")) as f:
        manifest = json.load(f)["cases"]
    return manifest, np.load(os.path.join(gdir, "golden_online_v1.npz"))


@pytest.fixture(scope="module")
def orc(oracle):
    lib = C.CDLL(ORACLE_SO)
    lib.online_oracle_pack.restype = C.c_int64
    lib.online_oracle_pack.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_size_t)]
    lib.online_oracle_unpack.restype = C.c_int64
    lib.online_oracle_unpack.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    lib.online_oracle_bound.restype = C.c_size_t
    lib.online_oracle_bound.argtypes = [C.c_int, C.c_uint32]
    return lib


def oracle_pack(lib, kind, x):
    x = np.ascontiguousarray(x, dtype=np.uint16)
    out = np.zeros(lib.online_oracle_bound(kind, x.size), np.uint8)
    nb = C.c_size_t(0)
    ret = lib.online_oracle_pack(kind, x.ctypes.data, x.size, out.ctypes.data, C.byref(nb))
    return out[: 2 * int(ret)].copy(), int(ret), int(nb.value)


def oracle_unpack(lib, kind, cont, n):
    buf = np.concatenate([np.ascontiguousarray(cont, dtype=np.uint8), np.zeros(64, np.uint8)])
    out = np.zeros(n + 16, np.uint16)
    ret = lib.online_oracle_unpack(kind, buf.ctypes.data, out.ctypes.data)
    return out[:n].copy(), int(ret)


def test_manifest_covers_every_coder_and_edge(golden_online):
    manifest, _ = golden_online
    assert {m["kind"] for m in manifest} == {0, 1, 2, 3, 4}
    assert {0, 1, 2, 8, 9, 10, 4113} <= {m["n"] for m in manifest}


def test_oracle_matches_golden(orc, golden_online):
    manifest, arrays = golden_online
    for m in manifest:
        x = arrays[m["name"] + "_in"]
        want = arrays[m["name"] + "_container"]
        got, ret, nbytes = oracle_pack(orc, m["kind"], x)
        assert ret == m["ret"], m
        assert nbytes in (2 * ret, 2 * ret - 1), m                 # exact bytes; the return value rounds up to elements
        assert np.array_equal(got, want), m                        # unwritten reference bytes are 0 on both sides
        back, dret = oracle_unpack(orc, m["kind"], want, x.size)
        assert dret == x.size and np.array_equal(back, x), m


def test_oracle_matches_compiled_reference(orc):
    if not os.path.exists(REF_SO):
        pytest.skip("oracle/_ref not built (reference sources absent)")
    ref = C.CDLL(REF_SO)
    ref.ref_online_pack.restype = C.c_int64
    ref.ref_online_pack.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_void_p]
    ref.ref_online_unpack.restype = C.c_int64
    ref.ref_online_unpack.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(99)
    for kind in range(5):
        for trial in range(120):
            n = int(rng.integers(0, 700)) if trial % 10 else int(rng.integers(100000, 140000))
            x = gen_fuzz(rng, n, 2, int(rng.integers(0, 16))) if trial % 3 else gen_walk(rng, n, 1, 2, int(rng.integers(1, 400)))
            x = np.ascontiguousarray(x)
            cap = 2 * n + n // 8 + 256
            a, b = np.zeros(cap, np.uint8), np.full(cap, 0xFF, np.uint8)
            ra = ref.ref_online_pack(kind, x.ctypes.data, n, a.ctypes.data)
            ref.ref_online_pack(kind, x.ctypes.data, n, b.ctypes.data)
            got, ret, _ = oracle_pack(orc, kind, x)
            assert ret == ra, (kind, n)
            defined = a[: 2 * ret] == b[: 2 * ret]
            assert np.array_equal(got[defined], a[: 2 * ret][defined]) and not got[~defined].any(), (kind, n)
            back, dret = oracle_unpack(orc, kind, a[: 2 * ret], n)
            rb = np.zeros(n + 16, np.uint16)
            assert ref.ref_online_unpack(kind, a.ctypes.data, rb.ctypes.data) == n == dret
            assert np.array_equal(back, rb[:n]) and np.array_equal(back, x), (kind, n)

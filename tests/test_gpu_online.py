"""GPU parity of the "online" u16 coders (sprintz_amd/csrc/online.hip; reference: cpp/Compress/online.hpp:395-445):
containers byte-exact with the golden ones minted from the compiled reference (and with the oracle on random and
long streams), decoders invert them; device forms and the single-call forms; the reference's own names through
the drop-in symbols are covered by tests/test_gpu_dropin.py."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from harness import gen_fuzz, gen_walk
from test_online_cpu import oracle_pack, oracle_unpack, orc, golden_online  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from sprintz_amd import _lib
    return _lib


def gpu_pack_host(lib, kind, x):
    x = np.ascontiguousarray(x, dtype=np.uint16)
    out = np.zeros(int(lib.online_bound(kind, x.size)) + 64, np.uint8)
    ret = lib.online_pack(kind, x.ctypes.data, x.size, out.ctypes.data)
    assert ret >= 0, lib.last_error()
    return out[: 2 * int(ret)].copy(), int(ret)


def gpu_unpack_host(lib, kind, cont, n):
    buf = np.concatenate([np.ascontiguousarray(cont, dtype=np.uint8), np.zeros(64, np.uint8)])
    out = np.zeros(n + 16, np.uint16)
    ret = lib.online_unpack(kind, buf.ctypes.data, out.ctypes.data)
    return out[:n].copy(), int(ret)


def test_golden_containers_single_call(lib, golden_online):
    manifest, arrays = golden_online
    for m in manifest[::3]:
        x = arrays[m["name"] + "_in"]
        want = arrays[m["name"] + "_container"]
        got, ret = gpu_pack_host(lib, m["kind"], x)
        assert ret == m["ret"], m
        assert np.array_equal(got, want), m
        back, dret = gpu_unpack_host(lib, m["kind"], want, x.size)
        assert dret == x.size and np.array_equal(back, x), m


@pytest.mark.parametrize("chain", ["default", "1", "0"])
@pytest.mark.parametrize("kind", [0, 1, 2, 3, 4])
def test_device_forms_on_long_streams(lib, orc, kind, chain, monkeypatch):
    """streams spanning many workgroups and scan tiles, incl. the reference suite's 1024*1024+7 elements.  chain: the dynamic-delta decoder's
    one-pass form (a chained scan over tiles of 8 192 blocks) from 128 tiles on (default), from the first tile on ("1"), never ("0")"""
    import torch
    if chain != "default":
        if kind > 1:
            pytest.skip("only the dynamic-delta decoder has two forms")
        monkeypatch.setenv("SPRINTZ_MI355X_ONLINE_CHAIN", chain)
    rng = np.random.default_rng(1000 + kind)
    for n in (2049, 16 * 2048 + 1, 300001, 1024 * 1024 + 7):
        for x in (gen_walk(rng, n, 1, 2, 7), gen_fuzz(rng, n, 2, int(rng.integers(0, 12))),
                  np.concatenate([np.zeros(n // 2, np.uint16), gen_walk(rng, n - n // 2, 1, 2, 300)])):
            want, wret, _ = oracle_pack(orc, kind, x)
            dx = torch.from_numpy(x.view(np.int16)).cuda()
            dest = torch.zeros(int(lib.online_bound(kind, n)) + 64, dtype=torch.uint8, device="cuda")
            tmp = torch.empty(int(lib.online_tmp_bytes(kind, n)) + 64, dtype=torch.uint8, device="cuda")
            ret = torch.zeros(1, dtype=torch.int64, device="cuda")
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            lib.check(lib.online_pack_device(kind, dx.data_ptr(), n, dest.data_ptr(), ret.data_ptr(), tmp.data_ptr(), st))
            r = int(ret.item())
            assert r == wret, (kind, n)
            assert np.array_equal(dest[: 2 * r].cpu().numpy(), want), (kind, n)
            out = torch.zeros(n + 16, dtype=torch.int16, device="cuda")
            lib.check(lib.online_unpack_device(kind, dest.data_ptr(), n, out.data_ptr(), ret.data_ptr(), tmp.data_ptr(), st))
            assert int(ret.item()) == n
            assert np.array_equal(out[:n].cpu().numpy().view(np.uint16), x), (kind, n)
            # a wrong length is reported, not decoded past the buffers
            lib.check(lib.online_unpack_device(kind, dest.data_ptr(), n - 1, out.data_ptr(), ret.data_ptr(), tmp.data_ptr(), st))
            assert int(ret.item()) == lib.E_CORRUPT


def test_decoder_reads_reference_containers(lib, golden_online):
    """every golden container (the reference's own bytes, unwritten padding as minted) decodes to its input"""
    manifest, arrays = golden_online
    for m in manifest[1::3]:
        x = arrays[m["name"] + "_in"]
        back, dret = gpu_unpack_host(lib, m["kind"], arrays[m["name"] + "_container"], x.size)
        assert dret == x.size and np.array_equal(back, x), m

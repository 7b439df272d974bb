"""Deterministic synthetic inputs of SURVEY.md 8(d): splitmix64 seeded per chunk (seed ^ chunk
index), identical in C (oracle/synth.c -- the specification is in its header), numpy and torch,
so that bench.py, tools/ and the tests see the same bytes on any machine and any torch version.

    synth_torch(kind, esz, nchunks, rows, ndims, device, seed=123, step=8, chunk0=0) -> [nchunks*rows*ndims]
    synth_numpy(...)                                                                  -> same values, on the host
kind: "uniform" | "walk" | "walkflat"
"""
import numpy as np

GOLDEN = 0x9E3779B97F4A7C15
M1, M2 = 0xBF58476D1CE4E5B9, 0x94D049BB133111EB
KINDS = {"uniform": 0, "walk": 1, "walkflat": 2}


def _s64(x):
    """a 64-bit constant as the signed value torch's int64 wants"""
    x &= (1 << 64) - 1
    return x - (1 << 64) if x >> 63 else x


def synth_numpy(kind, esz, nchunks, rows, ndims, seed=123, step=8, chunk0=0):
    w = 8 * esz
    with np.errstate(over="ignore"):
        c = (np.arange(nchunks, dtype=np.uint64) + np.uint64(chunk0)) ^ np.uint64(seed)
        i = np.arange(rows * ndims, dtype=np.uint64) + np.uint64(1)
        z = c[:, None] + i[None, :] * np.uint64(GOLDEN)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(M1)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(M2)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(32)).astype(np.int64).reshape(nchunks, rows, ndims)
    if kind == "uniform":
        x = u >> (32 - w)
    else:
        st = ((u * (2 * step + 1)) >> 32) - step
        if kind == "walkflat":
            st[:, (np.arange(rows) // 64) % 4 == 0] = 0
        st[:, 0] = u[:, 0] >> (32 - w)
        x = np.cumsum(st, axis=1)
    return (x & ((1 << w) - 1)).astype(np.uint8 if esz == 1 else np.uint16).reshape(-1)


def synth_torch(kind, esz, nchunks, rows, ndims, device, seed=123, step=8, chunk0=0, slab_elems=1 << 27):
    """the same values on `device`; uint8 / uint16 tensor.  Built in slabs of whole chunks so that the
    int64 temporaries stay bounded (an 800 000-chunk batch is 4 G elements)."""
    import torch
    w = 8 * esz
    out = torch.empty(nchunks * rows * ndims, dtype=torch.uint8 if esz == 1 else torch.uint16, device=device)
    per = rows * ndims
    slab = max(1, slab_elems // per)
    i = (torch.arange(per, dtype=torch.int64, device=device) + 1) * _s64(GOLDEN)       # wraps mod 2^64

    def lsr(z, k):
        return (z >> k) & ((1 << (64 - k)) - 1)

    flat = None
    if kind == "walkflat":
        flat = (torch.arange(rows, device=device) // 64) % 4 == 0
    for c0 in range(0, nchunks, slab):
        n = min(slab, nchunks - c0)
        c = (torch.arange(n, dtype=torch.int64, device=device) + (chunk0 + c0)) ^ _s64(seed)
        z = c[:, None] + i[None, :]
        z = (z ^ lsr(z, 30)) * _s64(M1)
        z = (z ^ lsr(z, 27)) * _s64(M2)
        z = z ^ lsr(z, 31)
        u = lsr(z, 32).view(n, rows, ndims)
        del z
        if kind == "uniform":
            x = u >> (32 - w)
        else:
            st = ((u * (2 * step + 1)) >> 32) - step
            if flat is not None:
                st[:, flat] = 0
            st[:, 0] = u[:, 0] >> (32 - w)
            x = torch.cumsum(st, dim=1)
            del st
        x = x & ((1 << w) - 1)
        if esz == 1:
            out[c0 * per:(c0 + n) * per] = x.to(torch.uint8).view(-1)
        else:
            out[c0 * per:(c0 + n) * per] = x.to(torch.int32).to(torch.uint16).view(-1)
        del x, u
    return out


def synth_cut_rows(kind, esz, nchunks, chunk_len, ndims, device, seed=123, step=8):
    """chunks of chunk_len elements that do NOT hold whole rows of ndims (cfg3 at 1 KB: 1024 elements, 80 columns): series of
    lcm(chunk_len, ndims) elements each -- a whole number of rows AND of chunks -- generated as synth_torch chunks and cut every
    chunk_len elements.  -> [nchunks * chunk_len]"""
    import math
    per = chunk_len * ndims // math.gcd(chunk_len, ndims)
    nseries = (nchunks * chunk_len + per - 1) // per
    return synth_torch(kind, esz, nseries, per // ndims, ndims, device, seed=seed, step=step)[: nchunks * chunk_len].contiguous()


def synth_c(kind, esz, nchunks, rows, ndims, seed=123, step=8, chunk0=0, lib_path=None):
    """oracle/synth.c through ctypes (test infrastructure)"""
    import ctypes as C
    import os
    lib_path = lib_path or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "liboracle.so")
    lib = C.CDLL(lib_path)
    lib.synth_fill.restype = None
    lib.synth_fill.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    out = np.zeros(nchunks * rows * ndims, np.uint8 if esz == 1 else np.uint16)
    lib.synth_fill(KINDS[kind], esz, seed, chunk0, nchunks, rows, ndims, step, out.ctypes.data)
    return out

#!/usr/bin/env python3
"""single-call latency (median us) of the drop-in symbols over shapes: which layouts the one-workgroup-per-chunk kernels cover"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from sprintz_amd import _lib
from synth import synth_numpy
for codec, esz, D, n in (("xff", 1, 1, 1024), ("xff", 1, 1, 10240), ("delta", 1, 1, 10240), ("xff", 2, 1, 5120), ("xff", 2, 2, 5120), ("xff", 1, 3, 9999 // 3 * 3),
                         ("xff", 2, 8, 5120), ("xff", 1, 8, 10240), ("delta", 1, 80, 10240), ("xff", 2, 32, 5120), ("xff", 2, 8, 20480), ("xff", 1, 8, 24576), ("xff", 2, 8, 32768)):
    x = synth_numpy("walk", esz, 1, n // D, D, seed=123, step=8 if esz == 2 else 2)
    cdst = np.zeros(n * 3 // 2 + 256, np.int16 if esz == 2 else np.int8)
    dst = np.zeros(n + 64, x.dtype)
    cfn, dfn = _lib.compress[(codec, esz)], _lib.decompress[(codec, esz)]
    r = cfn(x.ctypes.data, n, cdst.ctypes.data, D, 1)
    for _ in range(30):
        cfn(x.ctypes.data, n, cdst.ctypes.data, D, 1); dfn(cdst.ctypes.data, dst.ctypes.data)
    lc, ld = [], []
    for _ in range(300):
        t0 = time.perf_counter(); cfn(x.ctypes.data, n, cdst.ctypes.data, D, 1); lc.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); dfn(cdst.ctypes.data, dst.ctypes.data); ld.append(time.perf_counter() - t0)
    assert np.array_equal(dst[:n], x)
    lc.sort(); ld.sort()
    print(f"{codec:5s} u{8*esz:<2d} D={D:<3d} n={n:<6d} compress {lc[150]*1e6:7.1f} us   decompress {ld[150]*1e6:7.1f} us   ratio {n*esz/(r*esz):.2f}")

#!/usr/bin/env python3
"""The drop-in single-call symbols as lzbench drives them (one 10 KB chunk per call, host pointers): latency of one thread and
calls per second of 1 / 8 / 64 host threads, under each SPRINTZ_OPT_HOST_WAIT mode.  -> one JSON object on stdout.

    python tools/single_call_bench.py [--threads 1,8,64] [--modes 0,1,2]
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", default="1,8,64")
    ap.add_argument("--modes", default="0,1,2")
    a = ap.parse_args()
    from sprintz_amd import _lib
    from synth import synth_numpy
    chunk_len, ndims = 5120, 8
    raw1 = synth_numpy("walk", 2, 1, chunk_len // ndims, ndims, seed=123, step=8)
    cdst = np.zeros(chunk_len * 3 // 2 + 64, np.int16)
    dfn, cfn = _lib.decompress[("xff", 2)], _lib.compress[("xff", 2)]
    n = cfn(raw1.ctypes.data, chunk_len, cdst.ctypes.data, ndims, 1)
    one = cdst[: n + 8].copy()
    out = {}
    for mode in [int(m) for m in a.modes.split(",")]:
        _lib.check(_lib.set_option(_lib.OPT_HOST_WAIT, mode))
        dst = np.zeros(chunk_len + 64, np.uint16)
        for _ in range(50):
            dfn(one.ctypes.data, dst.ctypes.data)
            cfn(raw1.ctypes.data, chunk_len, cdst.ctypes.data, ndims, 1)
        lat_d, lat_c = [], []
        for _ in range(500):
            t0 = time.perf_counter(); r = dfn(one.ctypes.data, dst.ctypes.data); lat_d.append(time.perf_counter() - t0)
            assert r == chunk_len
            t0 = time.perf_counter(); cfn(raw1.ctypes.data, chunk_len, cdst.ctypes.data, ndims, 1); lat_c.append(time.perf_counter() - t0)
        assert np.array_equal(dst[:chunk_len], raw1)
        lat_d.sort(); lat_c.sort()
        res = {"decompress_us_median": round(lat_d[250] * 1e6, 1), "compress_us_median": round(lat_c[250] * 1e6, 1),
               "decompress_us_p10": round(lat_d[50] * 1e6, 1), "compress_us_p10": round(lat_c[50] * 1e6, 1)}
        for nt in [int(t) for t in a.threads.split(",")]:
            calls = 400 if nt < 32 else 100
            bufs = [(np.zeros(chunk_len + 64, np.uint16), np.zeros(chunk_len * 3 // 2 + 64, np.int16)) for _ in range(nt)]
            go = threading.Barrier(nt + 1)

            def work(k):
                d, c = bufs[k]
                dfn(one.ctypes.data, d.ctypes.data)
                go.wait()
                for _ in range(calls):
                    dfn(one.ctypes.data, d.ctypes.data)
                    cfn(raw1.ctypes.data, chunk_len, c.ctypes.data, ndims, 1)
            ths = [threading.Thread(target=work, args=(k,)) for k in range(nt)]
            [t.start() for t in ths]
            go.wait()
            t0 = time.perf_counter()
            [t.join() for t in ths]
            dt = time.perf_counter() - t0
            assert all(np.array_equal(b[0][:chunk_len], raw1) for b in bufs)
            res[f"threads_{nt}_calls_per_s"] = round(2 * nt * calls / dt)
        out[f"host_wait_{mode}"] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""round 6's delta kernels (mask 9) against rounds 1 - 5's (mask 0) over BATCH sizes and CHUNK sizes: uint8 x 80 columns, delta codec; ms"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import sprintz_amd
from sprintz_amd import _lib
from synth import synth_torch

dev = torch.device("cuda:0")
w = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
for _ in range(200):
    w.add_(1)
torch.cuda.synchronize()


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / reps)
    return best


D, esz = 80, 1
cases = [(128, n) for n in (2049, 4096, 8192, 16384, 41943)] + [(32, 160000), (64, 80000), (256, 20000), (312, 16000)]
for rows, n in cases:
    x = synth_torch("walk", esz, n, rows, D, dev, seed=5, step=8).view(torch.int8)
    cd = sprintz_amd.ChunkedCodec("delta", esz, D, rows * D, device=dev)
    res = {}
    for mask in (9, 0):
        _lib.check(_lib.set_option(_lib.OPT_BLK_KERNELS, mask))
        batch = cd.compress(x)
        out = torch.empty_like(x)
        te = timed(lambda: cd.compress(x))
        td = timed(lambda: cd.decompress(batch, out=out))
        assert torch.equal(out, x)
        res[mask] = (td, te)
    _lib.check(_lib.set_option(_lib.OPT_BLK_KERNELS, 9))
    flag = ("  <-- decode SLOWER" if res[9][0] > res[0][0] * 1.03 else "") + ("  <-- encode SLOWER" if res[9][1] > res[0][1] * 1.03 else "")
    print("chunk %5d B x %6d chunks  round 6: dec %.4f enc %.4f   before: dec %.4f enc %.4f%s" % (rows * D, n, res[9][0], res[9][1], res[0][0], res[0][1], flag), flush=True)

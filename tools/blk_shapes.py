#!/usr/bin/env python
"""round 6's delta kernels (encode_blk + decode_row: SPRINTZ_OPT_BLK_KERNELS = 9) against the kernels of rounds 1 - 5 (= 0) over element sizes and
column counts: the delta codec, general layout, 10 KB chunks, ~400 MB of walk data (steps in [-8, 8])"""
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


for esz, D in ((1, 16), (1, 32), (1, 48), (1, 64), (1, 80), (1, 128), (1, 256), (2, 8), (2, 16), (2, 24), (2, 32), (2, 64), (2, 128)):
    rows = 10240 // esz // D
    n = (400 << 20) // (rows * D * esz)
    x = synth_torch("walk", esz, n, rows, D, dev, seed=5, step=8)
    x = x.view(torch.int8 if esz == 1 else torch.int16)
    cd = sprintz_amd.ChunkedCodec("delta", esz, D, rows * D, device=dev)
    res = {}
    for mask in (9, 0):
        _lib.check(_lib.set_option(_lib.OPT_BLK_KERNELS, mask))
        batch = cd.compress(x)
        out = torch.empty_like(x)
        te = timed(lambda: cd.compress(x))
        td = timed(lambda: cd.decompress(batch, out=out))
        assert torch.equal(out, x)
        res[mask] = (td, te, batch.stream_bytes())
    _lib.check(_lib.set_option(_lib.OPT_BLK_KERNELS, 9))
    raw = x.numel() * esz
    fr = lambda t, sb: (raw + sb) / (t * 1e-3) / 8e12
    flag = ("  <-- decode SLOWER" if res[9][0] > res[0][0] * 1.03 else "") + ("  <-- encode SLOWER" if res[9][1] > res[0][1] * 1.03 else "")
    print("u%d x %3d  round 6: dec %.4f (%.3f) enc %.4f (%.3f)   before: dec %.4f (%.3f) enc %.4f (%.3f)%s" % (
        8 * esz, D, res[9][0], fr(res[9][0], res[9][2]), res[9][1], fr(res[9][1], res[9][2]), res[0][0], fr(res[0][0], res[0][2]), res[0][1], fr(res[0][1], res[0][2]), flag), flush=True)

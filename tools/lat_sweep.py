#!/usr/bin/env python3
"""decode_lat.h (one workgroup per chunk) against decode_fast.h (one lane per column) over small batch sizes of the headline
shape: where SPRINTZ_OPT_LAT_CHUNKS belongs.  -> markdown table, us per batched decompress (HIP events, data in HBM)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import sprintz_amd  # noqa: E402
from sprintz_amd import _lib  # noqa: E402
from synth import synth_torch  # noqa: E402


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


shapes = [("xff", 2, 8, 5120), ("delta", 1, 80, 10240), ("xff", 2, 32, 5120)]
sizes = [1, 64, 256, 640, 1250, 1536, 2048, 3072, 4096, 8192]
for codec, esz, D, chunk_len in shapes:
    print(f"\n{codec} u{8 * esz} D={D} chunk {chunk_len * esz} B\n| chunks | lat us | wide us |\n|---|---|---|")
    for n in sizes:
        x = synth_torch("walk", esz, n, chunk_len // D, D, "cuda:0", seed=123, step=8 if esz == 2 else 2)
        cd = sprintz_amd.ChunkedCodec(codec, esz, D, chunk_len, device="cuda:0")
        b = cd.compress(x)
        out = torch.empty(n * chunk_len, dtype=x.dtype, device="cuda:0")
        res = []
        for lat in (1 << 30, 0):
            _lib.check(_lib.set_option(_lib.OPT_LAT_CHUNKS, lat))
            cd.decompress_into(b.data, b.offsets, n, out)
            assert torch.equal(out.view(torch.uint8), x.view(torch.uint8))
            res.append(timeit(lambda: cd.decompress_into(b.data, b.offsets, n, out)))
        print(f"| {n} | {res[0]:.1f} | {res[1]:.1f} |")

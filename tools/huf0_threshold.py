#!/usr/bin/env python3
"""the Huff0 reader's two stream-kernel forms on either side of SPRINTZ_OPT_HUF0_BIG_BATCH: us per batch with the threshold above / below the batch
    python tools/huf0_threshold.py [chunks ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import sprintz_amd  # noqa: E402
from sprintz_amd import _lib  # noqa: E402
from synth import synth_torch  # noqa: E402

for n in [int(a) for a in sys.argv[1:]] or [10000, 20000, 30000, 40000, 60000]:
    x = synth_torch("walk", 2, n, 640, 8, "cuda:0", seed=123, step=8)
    cd = sprintz_amd.ChunkedCodec("xff", 2, 8, 5120, device="cuda:0")
    b = cd.compress(x)
    gb, gbo = sprintz_amd.huf0_compress(b)
    goo = torch.zeros(n + 1, dtype=torch.int64, device="cuda:0")
    goo[1:] = torch.cumsum(b.sizes.to(torch.int64), 0)
    st = sprintz_amd.huf0_decompress(gb, gbo, goo)
    res = []
    for thr in (1 << 30, 1):
        _lib.check(_lib.set_option(_lib.OPT_HUF0_BIG_BATCH, thr))
        for _ in range(3):
            sprintz_amd.huf0_decompress(gb, gbo, goo, out=st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            sprintz_amd.huf0_decompress(gb, gbo, goo, out=st)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    print(f"{n:7d} chunks: small form {res[0]:7.1f} us   big form {res[1]:7.1f} us")

#!/usr/bin/env python3
"""does bounding the OPEN output footprint of the headline decode help?  The 131 072 chunks as K back-to-back
launches of 131 072 / K chunks (K = 1, 2, 4, 8, 16): with K = 8 at most 168 MB of samples are being written at any time
(< the 256 MiB Infinity Cache).  Run once with the shipped library (nt stores) and once with a -DSPRINTZ_STORE_AUX=0 build
(plain stores): SPRINTZ_MI355X_LIB selects.  -> ms per whole batch."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import sprintz_amd  # noqa: E402
from sprintz_amd import _lib  # noqa: E402
from synth import synth_torch  # noqa: E402

n, chunk_len, D = 131072, 5120, 8
x = synth_torch("walk", 2, n, chunk_len // D, D, "cuda:0", seed=123, step=8)
cd = sprintz_amd.ChunkedCodec("xff", 2, D, chunk_len, device="cuda:0")
b = cd.compress(x)
out = torch.empty(n * chunk_len, dtype=torch.uint16, device="cuda:0")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(k):
    per = n // k
    for j in range(k):
        _lib.check(_lib.decompress_batch(_lib.CODEC_XFF, 2, b.data.data_ptr(), b.offsets.data_ptr() + 8 * j * per, per, chunk_len, D,
                                         out.data_ptr() + 2 * j * per * chunk_len, None, st))


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for _ in range(30):
    run(1)
print("lib:", os.environ.get("SPRINTZ_MI355X_LIB", "shipped"))
for k in (1, 2, 4, 8, 16, 1):
    out.zero_()
    run(k)
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int16), x.view(torch.int16)), k
    print(f"K = {k:2d} launches ({n // k * chunk_len * 2 / 1e6:7.1f} MB of output each): {timeit(lambda: run(k)):.4f} ms")

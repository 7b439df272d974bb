#!/usr/bin/env python3
"""phase durations of decode_lat.h on one chunk (a build with -DSPRINTZ_LAT_TIMING: SPRINTZ_MI355X_LIB=sprintz_amd/variants/lat_timing.so)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import sprintz_amd  # noqa: E402
from synth import synth_torch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
codec = sys.argv[2] if len(sys.argv) > 2 else "xff"
x = synth_torch("walk", 2, n, 640, 8, "cuda:0", seed=123, step=8)
cd = sprintz_amd.ChunkedCodec(codec, 2, 8, 5120, device="cuda:0")
b = cd.compress(x)
out = torch.empty(n * 5120, dtype=torch.uint16, device="cuda:0")
rets = torch.zeros(n, dtype=torch.int64, device="cuda:0")
for _ in range(5):
    cd.decompress_into(b.data, b.offsets, n, out, rets)
torch.cuda.synchronize()
r = rets.cpu().numpy()
names = ["load", "A|B|C pipeline", "D1+D2 sums", "D3 samples", "store"] if n != 2 else ["A end", "B1 round0", "B1 end", "C first", "C end"]
import numpy as np
for k, nm in enumerate(names):
    v = (r >> (12 * k)) & 4095
    print(f"{nm:14s} median {np.median(v) * 20:.0f} ns   max {v.max() * 20} ns")
print("total median", sum(np.median((r >> (12 * k)) & 4095) for k in range(5)) * 20, "ns")

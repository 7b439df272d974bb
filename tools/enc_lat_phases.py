#!/usr/bin/env python3
"""phase durations of encode_lat.h on one chunk (a -DSPRINTZ_LAT_TIMING build via SPRINTZ_MI355X_LIB)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import sprintz_amd
from synth import synth_torch
esz, D, n = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2, 8, 5120)
x = synth_torch("walk", esz, 4, n // D, D, "cuda:0", seed=123, step=8 if esz == 2 else 2)
cd = sprintz_amd.ChunkedCodec("xff", esz, D, n, device="cuda:0")
src = cd._padded_view(x); ws = cd.workspace(4)
for _ in range(5):
    cd.compress_to_slots(src, x.numel(), ws)
torch.cuda.synchronize()
r = ws["rets"].cpu().numpy()
names = ["load+zero", "deltas", "coef chain", "errors+widths", "RLE walk", "pack+tail", "store"]
tot = 0
for k, nm in enumerate(names):
    v = (r >> (9 * k)) & 511
    tot += np.median(v) * 20
    print(f"{nm:14s} median {np.median(v) * 20:.0f} ns")
print("total", tot, "ns")

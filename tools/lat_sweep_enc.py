#!/usr/bin/env python3
"""encode_lat.h (one workgroup per chunk) against encode_fast.h / encode_wide.h over batch sizes: us per compress_to_slots call"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import sprintz_amd
from sprintz_amd import _lib
from synth import synth_torch

def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for codec, esz, D, chunk_len in (("xff", 2, 8, 5120), ("xff", 2, 32, 5120), ("delta", 1, 8, 10240), ("xff", 1, 1, 1024)):
    print(f"\n{codec} u{8*esz} D={D} chunk {chunk_len*esz} B\n| chunks | lat us | wide us |\n|---|---|---|")
    for n in (256, 1024, 2048, 3072, 4096, 6144, 8192, 16384):
        x = synth_torch("walk", esz, n, chunk_len // D, D, "cuda:0", seed=123, step=8 if esz == 2 else 2)
        cd = sprintz_amd.ChunkedCodec(codec, esz, D, chunk_len, device="cuda:0")
        src = cd._padded_view(x); ws = cd.workspace(n)
        res = []
        for lat in (1 << 30, 0):
            _lib.check(_lib.set_option(_lib.OPT_LAT_CHUNKS, lat))
            res.append(timeit(lambda: cd.compress_to_slots(src, x.numel(), ws)))
        print(f"| {n} | {res[0]:.1f} | {res[1]:.1f} |")

#!/usr/bin/env python
"""time the stand-alone transforms' decode (delta, double delta) on the bench's stream, for the library SPRINTZ_MI355X_LIB names:
tools/transforms_ab.py [reps [kind [size]]]  ->  kind, 64 Mi samples: ms, frac; 512 Mi samples: ms, frac (round trip checked)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import sprintz_amd
from synth import synth_torch

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
only_kind = sys.argv[2] if len(sys.argv) > 2 else ""      # "delta" / "doubledelta": that kind only
only_size = sys.argv[3] if len(sys.argv) > 3 else ""      # "64Mi" / "512Mi"
D, rows = 8, (64 << 20) // 8
x = synth_torch("walk", 2, 1, rows, D, dev, seed=123, step=8).reshape(-1)


def timed(fn, reps):
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


# warm the clock
w = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
for _ in range(200):
    w.add_(1)
torch.cuda.synchronize()
for name in ("64Mi", "512Mi"):
    if only_size and name != only_size:
        continue
    src = x if name == "64Mi" else x.repeat(8)
    for kind in ("delta", "doubledelta"):
        if only_kind and kind != only_kind:
            continue
        y, back = torch.empty_like(src), torch.empty_like(src)
        sprintz_amd.transform_device(kind, src, D, out=y)
        td = timed(lambda: sprintz_amd.transform_device(kind, y, D, inverse=True, out=back), reps if name == "64Mi" else max(2, reps // 5))
        ok = torch.equal(back.view(torch.int16), src.view(torch.int16))
        nb = src.numel() * 2
        print("%s %-11s %s dec %.4f ms frac %.3f %s" % (os.environ.get("SPRINTZ_MI355X_LIB", "default")[-24:], kind, name, td, 2 * nb / (td * 1e-3) / 8e12, "ok" if ok else "MISMATCH"), flush=True)
        del y, back

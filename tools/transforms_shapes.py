#!/usr/bin/env python
"""the transforms' decode, one pass against two (SPRINTZ_MI355X_TRANSFORM_CHAIN=0), over element sizes and column counts: 128 MiB streams"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import sprintz_amd

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


g = torch.Generator(device="cuda")
g.manual_seed(1)
nbytes = 128 << 20
SHAPES = os.environ.get("SHAPES")      # "2x3,2x5": element bytes x columns
shapes = [tuple(int(v) for v in t.split("x")) for t in SHAPES.split(",")] if SHAPES else ((1, 1), (1, 2), (1, 4), (1, 8), (1, 16), (1, 80), (1, 128), (2, 1), (2, 2), (2, 4), (2, 8), (2, 24), (2, 64))
for esz, D in shapes:
    n = nbytes // esz // D * D
    x = torch.randint(0, 1 << (8 * esz), (n,), generator=g, device="cuda", dtype=torch.int32).to(torch.uint8 if esz == 1 else torch.uint16)
    for kind in ("delta", "doubledelta"):
        y = sprintz_amd.transform_device(kind, x, D)
        back = torch.empty_like(x)
        res = []
        for chain in ("", "0"):
            if chain:
                os.environ["SPRINTZ_MI355X_TRANSFORM_CHAIN"] = chain
            else:
                os.environ.pop("SPRINTZ_MI355X_TRANSFORM_CHAIN", None)
            t = timed(lambda: sprintz_amd.transform_device(kind, y, D, inverse=True, out=back))
            assert torch.equal(back, x)
            res.append(t)
        os.environ.pop("SPRINTZ_MI355X_TRANSFORM_CHAIN", None)
        print("u%d x %3d %-11s one pass %.4f ms (%.3f)   two passes %.4f ms (%.3f)  %s" % (8 * esz, D, kind, res[0], 2 * n * esz / res[0] / 8e9, res[1], 2 * n * esz / res[1] / 8e9,
                                                                                          "" if res[0] <= res[1] * 1.02 else "<-- SLOWER"), flush=True)

#!/usr/bin/env python3
"""Could the FIRE encoder go block-parallel like the delta encoder (csrc/encode_blk.h)?  ("one more structural attempt")

Runs HERE (CPU, numpy).  In the general 16-bit layout a column's only recurrence is its COUNTER: block b's coefficient is
coef_b = int16((counter_b >> 13) << 12) and counter_{b+1} = counter_b + (int16(sum over the odd rows of sign(err) * prev_delta) >> 2)
(sprintz_xff_rle.cpp:217, :240-241, :273-275), everything else is element-wise GIVEN coef_b.  A block-parallel encoder would guess the coefficients,
compute every block's gradient in parallel, scan the counters and repeat until the guess is self-consistent -- one iteration per coefficient CHANGE
along the chunk (the first wrong block is right after every pass).  This tool replays the counters on the bench's data and prints how many changes
a chunk's columns see and which coefficient values occur: the price of that iteration, and of the alternative that precomputes the gradients for a
fixed candidate set of coefficients and runs a 3-instruction recurrence over them."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
from synth import synth_numpy                   # noqa: E402


def i16(x):
    return ((x + 32768) & 0xffff) - 32768


def replay(x):
    """x: [chunks, rows, D] uint16 -> coefficient per (chunk, block, column) as the reference encoder's counters give it"""
    c, rows, D = x.shape
    nb = rows // 8
    xs = x.astype(np.int64)
    prev = np.zeros((c, D), np.int64)
    pdelta = np.zeros((c, D), np.int64)
    ctr = np.zeros((c, D), np.int64)
    coefs = np.zeros((c, nb, D), np.int64)
    for b in range(nb):
        coef = i16((ctr >> 13) << 12)
        coefs[:, b] = coef
        grad = np.zeros((c, D), np.int64)
        for i in range(8):
            cur = xs[:, 8 * b + i]
            delta = i16(cur - prev)
            pred = i16((pdelta * coef) >> 16)
            err = i16(delta - pred)
            if i & 1:
                grad = i16(grad + np.sign(err) * pdelta)
            prev, pdelta = cur, delta
        ctr = ctr + (grad >> 2)
    return coefs


def main():
    n, rows, D = 256, 640, 8
    for kind, step in (("walk", 8), ("walk", 300), ("uniform", 0)):
        x = np.ascontiguousarray(synth_numpy(kind, 2, n, rows, D, seed=123, step=step)).view(np.uint16).reshape(n, rows, D)
        co = replay(x)
        changes = (np.diff(co, axis=1) != 0).sum(axis=1)              # per (chunk, column)
        per_chunk = changes.max(axis=1)                               # passes a chunk needs = its worst column's changes (+ 1)
        vals, cnts = np.unique(co, return_counts=True)
        top = sorted(zip(cnts.tolist(), vals.tolist()), reverse=True)[:6]
        inside = ((co == 0) | (co == -4096)).all(axis=(1, 2)).mean()
        inside4 = np.isin(co, (-8192, -4096, 0, 4096)).all(axis=(1, 2)).mean()
        print(f"{kind:8s} step {step:4d}: coefficient changes per (chunk, column): mean {changes.mean():6.2f}, per chunk (worst column): mean {per_chunk.mean():6.2f}, "
              f"p90 {np.percentile(per_chunk, 90):5.0f}, max {per_chunk.max()}")
        print(f"{'':20s}blocks by coefficient (count, value): {top}")
        print(f"{'':20s}chunks whose every coefficient is in {{0, -4096}}: {inside:.3f}; in {{-8192, -4096, 0, 4096}}: {inside4:.3f}")


if __name__ == "__main__":
    main()

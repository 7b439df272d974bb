#!/usr/bin/env python3
"""How often would a two-symbols-per-look-up Huff0 table (zstd's "X2" decoder) deliver its second symbol on Sprintz streams?

Runs HERE (CPU, oracle only): encodes chunks of the bench's headline data with the oracle, builds the length-limited code the Huff0 writer's
specification uses (oracle huf_oracle_lengths, max 11 bits) over a 64-chunk segment, and walks the symbol sequence greedily: a look-up of T
bits yields two symbols when their code lengths sum to <= T.  Printed: entropy, mean code length, symbols per look-up for T = 11, 12.
DESIGN.md 4.4b quotes the result (walk +-8, uint16, 8 columns: 7.4 bits a symbol, 1.08 / 1.12 symbols per look-up)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import harness                                  # noqa: E402  (test infrastructure: this tool is an analysis aid, not product code)
from synth import synth_numpy                   # noqa: E402


def main():
    o = harness.Oracle()
    n, rows, D = 64, 640, 8
    x = np.ascontiguousarray(synth_numpy("walk", 2, n, rows, D, seed=123, step=8)).view(np.uint16).reshape(n, rows * D)
    allb = np.concatenate([o.compress("xff", x[c], D)[0] for c in range(n)])
    cnt = np.bincount(allb, minlength=256).astype(np.uint32)
    lens = o.huf_lengths(cnt)
    p = cnt / cnt.sum()
    print(f"{len(allb)} stream bytes of {n} chunks: entropy {-(p[p > 0] * np.log2(p[p > 0])).sum():.3f} bits, mean code length {(p * lens).sum():.3f}")
    print("symbols per code length 0..12:", np.bincount(lens, minlength=13).tolist())
    l = lens[allb].astype(int)
    for T in (11, 12):
        i = look = 0
        while i < len(l) - 1:
            i += 2 if l[i] + l[i + 1] <= T else 1
            look += 1
        print(f"table log {T}: P(adjacent pair fits) {((l[:-1] + l[1:]) <= T).mean():.3f}, symbols per look-up {len(l) / look:.3f}")


if __name__ == "__main__":
    main()

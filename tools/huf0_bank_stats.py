#!/usr/bin/env python3
"""What do the big-batch Huff0 stream kernel's table look-ups cost in LDS bank conflicts, and can a layout remove them?  

Runs HERE (CPU, oracle only).  The kernel (csrc/huf0.hip: huf0_stream_kernel<true, ...>) keeps ONE 2^11-entry decode table per workgroup and every
lane looks up the next 11 bits of ITS stream: 64 unrelated indices a wave-instruction.  A ds_read of <= 4 bytes is served 32 lanes at a time over 32
banks of 4 bytes (MI355X_MICROARCH.md, LDS): a group of lanes costs as many cycles as the most loaded bank has DISTINCT dwords.  This tool takes the
bench's own streams (oracle-encoded walk data, the 64-chunk segment code of the Huff0 writer's specification), collects the 11-bit windows at real
symbol boundaries, draws waves of 64 of them and prices:
  uniform      : 64 uniformly random indices (what no layout can beat without replication)
  as built     : entry i (2 bytes) at byte 2 i
  padded       : entry i at i + i / 32 entries (the verdict's suggestion)
  hashed       : entry i at i ^ (i >> 5) (spreads aligned runs of equal entries)
  4-byte       : entry i (4 bytes) at byte 4 i -- one entry a bank word
  first level  : a 64-entry table of the 6-bit prefixes replicated over the banks (conflict-free), the full table for the misses: cycles =
                 1 + the conflicts among the lanes that miss; and the fraction of wave-instructions in which NO lane misses
Printed: mean LDS cycles per 32-lane group, per layout."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import harness                                  # noqa: E402  (test infrastructure: an analysis aid, not product code)
from synth import synth_numpy                   # noqa: E402

TL = 11


def canonical_codes(lens):
    """Huff0's canonical code (HUF_readDTableX1 fill order): per weight ascending symbols, the longest codes lowest -> table start index per symbol"""
    tl = int(lens.max())
    start = np.zeros(256, np.int64)
    at = 0
    for w in range(1, tl + 1):                   # weight w <-> length tl + 1 - w
        for s in range(256):
            if lens[s] and tl + 1 - lens[s] == w:
                start[s] = at
                at += 1 << (w - 1)
    return start, tl


def cycles(addr_dword):
    """addr_dword: [waves, 32] dword addresses of one 32-lane group -> mean cycles (max over banks of distinct dwords)"""
    out = np.zeros(addr_dword.shape[0], np.int64)
    for i, row in enumerate(addr_dword):
        u = np.unique(row)
        out[i] = np.bincount(u % 32, minlength=32).max()
    return out.mean()


def main():
    o = harness.Oracle()
    n, rows, D = 64, 640, 8
    x = np.ascontiguousarray(synth_numpy("walk", 2, n, rows, D, seed=123, step=8)).view(np.uint16).reshape(n, rows * D)
    allb = np.concatenate([o.compress("xff", x[c], D)[0] for c in range(n)])
    cnt = np.bincount(allb, minlength=256).astype(np.uint32)
    lens = o.huf_lengths(cnt).astype(np.int64)
    start, tl = canonical_codes(lens)
    assert tl <= TL
    rng = np.random.default_rng(1)
    # the 11-bit window at a symbol boundary = the symbol's code followed by the next symbols' codes: index = start[s] .. + 2^(tl-len) (following bits)
    # (model: the bits behind a code are the next codes' bits -- drawn from the stream's own symbol sequence)
    m = 200000
    pos = rng.integers(0, len(allb) - 4, m)
    span = (1 << (tl - lens[allb[pos]]))
    idx = start[allb[pos]] + (rng.integers(0, 1 << 30, m) % span)      # following bits ~ uniform within the code's span (a good model for a prefix code's tail)
    idx <<= (TL - tl)
    waves = 4000
    pick = rng.integers(0, m, (waves, 32))
    I = idx[pick]
    U = rng.integers(0, 1 << TL, (waves, 32))
    print(f"{len(allb)} stream bytes, code lengths {np.bincount(lens, minlength=13).tolist()} (symbols per length), table log {tl}")
    print(f"mean code length {(cnt / cnt.sum() * lens).sum():.3f} bits; P(length <= 6) by symbol frequency {(cnt[lens <= 6][lens[lens <= 6] > 0].sum() / cnt.sum()):.3f}, <= 8: {(cnt[(lens <= 8) & (lens > 0)].sum() / cnt.sum()):.3f}")
    rowsout = []
    rowsout.append(("uniform random indices, 2-byte entries", cycles(U >> 1)))
    rowsout.append(("as built: entry i at byte 2 i", cycles(I >> 1)))
    rowsout.append(("padded: entry i at i + i / 32", cycles((I + I // 32) >> 1)))
    rowsout.append(("hashed: entry i at i ^ (i >> 5)", cycles((I ^ (I >> 5)) >> 1)))
    rowsout.append(("4-byte entries: entry i at byte 4 i", cycles(I)))
    for bits in (6, 8):
        short = lens[allb[pos]] <= bits
        miss = ~short[pick]
        cyc = np.zeros(waves)
        for w in range(waves):
            mi = I[w][miss[w]]
            cyc[w] = 1 + (np.bincount(np.unique(mi >> 1) % 32, minlength=32).max() if mi.size else 0)
        rowsout.append((f"first level of {1 << bits} prefixes replicated over the banks + the full table for the misses", cyc.mean()))
        rowsout.append((f"   ... fraction of 32-lane groups with no miss at all", float((miss.sum(axis=1) == 0).mean())))
    for name, v in rowsout:
        print(f"{name:95s} {v:7.3f}")
    print("(a conflict-free look-up is 1.0; the kernel's measured SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 435 M / 670 M = 0.65, i.e. 2.9 cycles a group)")


if __name__ == "__main__":
    main()

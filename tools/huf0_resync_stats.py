#!/usr/bin/env python3
"""Would SELF-SYNCHRONISING sub-stream decoding shorten the Huff0 stage's per-stream chain on Sprintz streams?

A Huff0 stream is one serial chain of ~900 table look-ups.  A decoder started at an ARBITRARY bit of the stream decodes garbage until
its cursor happens to land on a true code boundary, from where on it is right; split a stream into K spans, start one decoder per span
boundary, let each run over its span plus an overlap, keep what was decoded after the sync point.  That only pays if the sync point
comes soon.  This tool prices it on the CPU (oracle only, nothing of the product): the bench's headline chunks (walk +-8, uint16 x 8
columns) are encoded with the oracle, the length-limited code of the Huff0 writer's specification (oracle huf_oracle_lengths, <= 11
bits, one per 64-chunk segment) is built as a canonical code, the chunks' byte sequences are laid out as the bit sequence a decoder
walks, and decoders are started at random bit offsets: printed are the distribution of the number of symbols (and bits) decoded before
the cursor first meets a true boundary, and how many starts never meet one within 512 symbols.
DESIGN.md 4.6 quotes the result."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import harness                                  # noqa: E402  (test infrastructure: this tool is an analysis aid, not product code)
from synth import synth_numpy                   # noqa: E402


def canonical(lens):
    """symbol -> (code, length) of the canonical code: shorter codes first, symbols ascending within a length"""
    order = sorted((int(l), s) for s, l in enumerate(lens) if l)
    code, prev, out = 0, order[0][0], {}
    for l, s in order:
        code <<= l - prev
        prev = l
        out[s] = (code, l)
        code += 1
    return out


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "walk"
    o = harness.Oracle()
    n, rows, D = 64, 640, 8
    x = np.ascontiguousarray(synth_numpy(kind, 2, n, rows, D, seed=123, step=8)).view(np.uint16).reshape(n, rows * D)
    streams = [o.compress("xff", x[c], D)[0] for c in range(n)]
    allb = np.concatenate(streams)
    lens = o.huf_lengths(np.bincount(allb, minlength=256).astype(np.uint32))
    cw = canonical(lens)
    maxl = int(max(lens))
    table = np.zeros(1 << maxl, np.int32)                         # look-ahead of maxl bits -> code length
    for s, (c, l) in cw.items():
        table[c << (maxl - l):(c + 1) << (maxl - l)] = l
    rng = np.random.default_rng(1)
    syms_to_sync, bits_to_sync, never = [], [], 0
    trials_per_chunk = 400
    for st in streams:
        l = lens[st].astype(np.int64)
        pos = np.concatenate([[0], np.cumsum(l)])                 # true boundaries
        nbits = int(pos[-1])
        bits = np.zeros(nbits + maxl, np.uint8)
        for i, s in enumerate(st):                                # (3 600 symbols a chunk: a plain loop is fine)
            c, ll = cw[int(s)]
            p = int(pos[i])
            for k in range(ll):
                bits[p + k] = (c >> (ll - 1 - k)) & 1
        is_boundary = np.zeros(nbits + maxl + 1, bool)
        is_boundary[pos] = True
        weights = 1 << np.arange(maxl - 1, -1, -1)
        for _ in range(trials_per_chunk):
            p0 = int(rng.integers(0, max(1, nbits - 600 * maxl)))
            if is_boundary[p0]:
                continue                                          # (a start that happens to be right says nothing)
            p, k = p0, 0
            while k < 512 and p < nbits and not is_boundary[p]:
                p += int(table[int((bits[p:p + maxl] * weights).sum())])
                k += 1
            if p < nbits and is_boundary[p]:
                syms_to_sync.append(k)
                bits_to_sync.append(p - p0)
            else:
                never += 1
    s = np.array(syms_to_sync)
    b = np.array(bits_to_sync)
    p = np.bincount(allb, minlength=256) / allb.size
    print(f"data {kind}: {allb.size} stream bytes, mean code length {(p * lens).sum():.2f} bits, code lengths used {sorted(set(int(v) for v in lens if v))}")
    print(f"{s.size + never} wrong starts: {never} ({100.0 * never / (s.size + never):.1f} %) not synchronised within 512 symbols")
    for q in (50, 75, 90, 95, 99):
        print(f"  {q:2d} % synchronised within {int(np.percentile(s, q)):4d} symbols / {int(np.percentile(b, q)):5d} bits")
    print(f"  mean {s.mean():.1f} symbols")
    for K in (4, 8, 16):
        span = 900 / K
        ov = np.percentile(s, 90)
        print(f"  {K:2d} sub-sequences of a 900-symbol stream: span {span:.0f} symbols + overlap to the 90th percentile {ov:.0f} -> "
              f"chain {span + ov:.0f} symbols instead of 900 ({900 / (span + ov):.2f}x shorter), decoded work {1 + ov / span:.2f}x; "
              f"{100 - 90} % of the spans need a second pass")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Mint golden vectors for the Huff0 wire format from the system libzstd (HUF_compress, zstd
1.4.x) -- the stand-in for the un-vendored Huff0 the paper uses.  TEST
INFRASTRUCTURE ONLY; run where libzstd.so.1 exists:

    python oracle/gen_golden_huf0.py

Writes tests/golden/golden_huf0_v1.npz (+ .json): for every case the plain bytes and the block
HUF_compress wrote for them (stored / one-byte blocks under HUF_decompress's conventions).
The plain bytes are Sprintz streams of deterministic inputs (what the stage is applied to) and
synthetic symbol distributions that reach the format's corners (4-bit and FSE weight headers,
few symbols, 256 symbols, table log 11, tiny and 128 KB blocks).  Only data is stored."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from harness import Oracle, Zstd, gen_walk  # noqa: E402


def main():
    z, o = Zstd(), Oracle()
    rng = np.random.default_rng(20240931)
    plains = []
    for esz, D, codec in ((2, 8, "xff"), (1, 1, "delta"), (1, 80, "delta"), (2, 32, "xff"), (2, 2, "xff")):
        data = gen_walk(rng, 24 * 5120, D, esz, 8, flat_every=4)
        for st in o.compress_chunks(codec, data, 5120, D):
            plains.append(("sprintz_%s_%d_%d" % (codec, 8 * esz, D), np.ascontiguousarray(st)))
    for n in (11, 12, 13, 50, 255, 256, 1000, 4096, 65536, 131072):
        for k in (1, 2, 3, 5, 16, 64, 128, 129, 200, 256):
            for skew in (0.0, 1.0, 3.0):
                if n >= 65536 and (k not in (16, 200, 256) or skew != 1.0):
                    continue
                p = 1.0 / np.arange(1, k + 1) ** skew
                s = rng.choice(k, n, p=p / p.sum()).astype(np.uint8)
                plains.append(("zipf_n%d_k%d_s%g" % (n, k, skew), s if (n + k) % 2 else (s * 7 + 3).astype(np.uint8)))
    log12 = []
    # HUF_compress2 with huffLog 12, the format's largest table.  (With a dominant symbol it emits a weight of
    # 12, which the library's own HUF_readStats rejects -- and so do the oracle and the kernel; those are left out.)
    for n in (12000, 20000):
        for skew in (1.0, 1.2, 1.4):
            p = 1.0 / np.arange(1, 257) ** skew
            cand = rng.choice(256, n, p=p / p.sum()).astype(np.uint8)
            if z.huf_decompress(z.huf_compress(cand, 12), n)[1] == n:
                log12.append(("log12_n%d_s%g" % (n, skew), cand))
    arrays, manifest, kinds = {}, [], {"stored": 0, "rle": 0, "fse": 0, "nibbles": 0}
    nlog12 = 0
    for i, (name, s) in enumerate(plains + log12):
        blk = z.huf_compress(s, 12 if name.startswith("log12") else None)
        nlog12 += name.startswith("log12") and o.huf0_table_log(blk) == 12
        back, r = o.huf0_decompress(blk, s.size)
        assert r == s.size and np.array_equal(back, s), name
        kind = "stored" if blk.size == s.size else "rle" if blk.size == 1 else "fse" if blk[0] < 128 else "nibbles"
        kinds[kind] += 1
        arrays["p%04d" % i] = s
        arrays["b%04d" % i] = blk
        manifest.append({"idx": i, "name": name, "n": int(s.size), "block": int(blk.size), "kind": kind})
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "golden_huf0_v1.npz"), **arrays)
    with open(os.path.join(ROOT, "tests", "golden", "golden_huf0_v1.json"), "w") as f:
        json.dump({"version": 1, "zstd": z.version, "cases": manifest}, f, indent=0)
    assert nlog12 >= 3, nlog12
    print(len(manifest), "cases", kinds, "table log 12:", nlog12)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Mint golden vectors for the reference's non-RLE codecs (compress_rowmajor_{8b,16b},
compress_rowmajor_delta_{8b,16b}: sprintz_delta.cpp:64-1391) from the COMPILED REFERENCE.
TEST INFRASTRUCTURE ONLY; run in the build container after `make -C oracle ref`:

    python oracle/gen_golden_norle.py

Writes tests/golden/golden_norle_v1.npz (+ .json): input, exact stream (length from the
oracle's framing walk, cross-checked against the element count the reference returns and
against a second run into a differently poisoned buffer), return value.  Only data is stored."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from harness import DTYPES, Oracle, Reference  # noqa: E402


def main():
    ref, orc = Reference(), Oracle()
    rng = np.random.default_rng(20241001)
    arrays, manifest = {}, []
    idx = 0
    for esz in (1, 2):
        top = 1 << (8 * esz)
        for raw in ((0, 1, 2) if esz == 1 else (0, 1)):      # 2: compress8b_rowmajor_xff (sprintz_xff.cpp), 8-bit only
            for D in (1, 2, 3, 4, 5, 8, 17, 33, 80):
                for n in (1, 17, 127, 128, 129, 16 * D, 16 * D + 1, 48 * D + 5, 1000, 4113):
                    for kind in ("fuzz", "walk", "walk_zero", "small"):
                        if kind == "fuzz":
                            x = rng.integers(0, top, n)
                        elif kind == "small":
                            x = rng.integers(0, 13, n)
                        else:
                            x = np.cumsum(rng.integers(-3, 4, n)) % top
                            if kind == "walk_zero":
                                x[n // 3: 2 * n // 3] = 0
                        x = x.astype(DTYPES[esz])
                        buf, ret = ref.compress_norle_raw(raw, x, D)
                        so, ro = orc.compress_norle(raw, x, D)
                        nb = len(so)
                        assert ro == ret and nb // esz == ret and np.array_equal(buf[:nb], so), (esz, raw, D, n, kind)
                        back, bret = ref.decompress_norle(raw, so, esz)
                        assert bret == n and np.array_equal(back, x), (esz, raw, D, n, kind)
                        name = f"n{idx:04d}"
                        arrays[name + "_in"] = x
                        arrays[name + "_stream"] = so
                        manifest.append({"name": name, "raw": raw, "esz": esz, "ndims": D, "n": n, "kind": kind, "ret": ret})
                        idx += 1
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "golden_norle_v1.npz"), **arrays)
    with open(os.path.join(ROOT, "tests", "golden", "golden_norle_v1.json"), "w") as f:
        json.dump({"version": 1, "cases": manifest}, f, indent=0)
    print(len(manifest), "cases")


if __name__ == "__main__":
    main()

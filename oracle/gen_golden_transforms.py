#!/usr/bin/env python3
"""Mint golden vectors for the stand-alone transforms (delta.h:17-68, predict.h:15-30) from the COMPILED
REFERENCE.  TEST INFRASTRUCTURE ONLY; run in the build container after `make -C oracle ref`:

    python oracle/gen_golden_transforms.py

Writes tests/golden/golden_transforms_v2.npz (+ .json): input, the container the reference's
encode_{delta,doubledelta,xff}_rowmajor_{8b,16b} wrote (6-byte header + len elements), its return
value, and what its decoder returns for that container.  Only data is stored."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from harness import DTYPES, Reference  # noqa: E402


def main():
    ref = Reference()
    rng = np.random.default_rng(20240930)
    arrays, manifest = {}, []
    idx = 0
    for esz in (1, 2):
        top = 1 << (8 * esz)
        for kind in (0, 1, 2):
            for D in (1, 2, 3, 8, 17, 32, 33, 80, 200):
                for n in (1, D, D + 1, 2 * D + 1, 8 * D, 8 * D + 3, 16 * D + 5, 1000, 4113) + ((40 * D + 7, 20000) if kind == 2 else ()):
                    x = (rng.integers(0, top, n) if (idx % 2) else np.cumsum(rng.integers(-5, 6, n)) % top).astype(DTYPES[esz])
                    cont, ret = ref.transform_encode(kind, x, D)
                    back, dret = ref.transform_decode(kind, cont, esz)
                    assert dret == n and np.array_equal(back, x), (esz, kind, D, n)
                    name = f"t{idx:04d}"
                    arrays[name + "_in"] = x
                    arrays[name + "_container"] = cont
                    manifest.append({"name": name, "kind": kind, "esz": esz, "ndims": D, "n": n, "ret": ret})
                    idx += 1
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "golden_transforms_v2.npz"), **arrays)
    with open(os.path.join(ROOT, "tests", "golden", "golden_transforms_v2.json"), "w") as f:
        json.dump({"version": 2, "cases": manifest}, f, indent=0)
    print(len(manifest), "cases")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Mint golden vectors for the reference's *_rowmajor_*_rle_* family (general row-major
layout for EVERY ndims) and its query path, from the COMPILED REFERENCE.

TEST INFRASTRUCTURE ONLY.  Run in the build container after `make -C oracle ref`:

    python oracle/gen_golden_rowmajor.py

Writes tests/golden/golden_rowmajor_v1.npz (+ .json manifest).  Per case: the input, the
exact stream compress_rowmajor_{delta,xff}_rle_{8b,16b} produced (sprintz_delta.h:49,
sprintz_xff.h:45-55 -- what the reference's query tests feed query_rowmajor_*,
test/test_query.cpp:59-120), and whether the reference's own
query_rowmajor_*(materialize=true) reproduced the input (it does not for xff 16b: that
test is commented out upstream, test/test_query.cpp:218-241).  Stream length: the bytes two
runs into differently poisoned buffers agree on, cross-checked against the element count
the reference returns (exact for 8-bit, floor for 16-bit).  Only data is stored.
"""
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
    rng = np.random.default_rng(20240929)
    arrays, manifest = {}, []
    idx = 0
    for esz in (1, 2):
        top = 1 << (8 * esz)
        for codec in ("delta", "xff"):
            for D in (1, 2, 3, 4, 5, 8, 17, 80):
                for n in (1, 17, 127, 128, 129, 16 * D, 16 * D + 1, 1000, 4113):
                    for kind in ("fuzz", "walk", "walk_flat"):
                        if kind == "fuzz":
                            x = rng.integers(0, top, n)
                        else:
                            x = np.cumsum(rng.integers(-3, 4, n)) % top
                            if kind == "walk_flat":
                                x[n // 3: 2 * n // 3] = 7
                        x = x.astype(DTYPES[esz])
                        buf, ret = ref.compress_rowmajor_raw(codec, x, D)
                        so, ro = orc.compress_rowmajor(codec, x, D)          # framing walk gives the exact length
                        nb = len(so)
                        assert ro == ret and nb // esz == ret and np.array_equal(buf[:nb], so), (esz, codec, D, n, kind)
                        d, qret = ref.query(codec, buf[:nb], esz, n, 0, True, D)
                        ref_ok = bool(qret == n and np.array_equal(d, x))
                        name = f"c{idx:04d}"
                        arrays[name + "_in"] = x
                        arrays[name + "_stream"] = buf[:nb].copy()
                        manifest.append({"name": name, "codec": codec, "esz": esz, "ndims": D, "n": n, "kind": kind,
                                         "ret": ret, "ref_query_materialize_ok": ref_ok})
                        idx += 1
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "golden_rowmajor_v1.npz"), **arrays)
    with open(os.path.join(ROOT, "tests", "golden", "golden_rowmajor_v1.json"), "w") as f:
        json.dump({"version": 1, "cases": manifest}, f, indent=0)
    bad = [m for m in manifest if not m["ref_query_materialize_ok"]]
    print(len(manifest), "cases;", len(bad), "where the reference's own query(materialize) != input:",
          sorted({(m["codec"], m["esz"]) for m in bad}))


if __name__ == "__main__":
    main()

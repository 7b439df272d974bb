#!/usr/bin/env python3
"""Streams of 513 .. 2 047 columns minted by the compiled REFERENCE (make -C oracle ref): the reference's own tests stop at 129 columns
(test/compress_testing.hpp:20-21), its encoder and decoder take any u16 ndims that fits a group, and the kernels of csrc/any_ndims.hip
exist for exactly these widths.  -> tests/golden/golden_wide_v1.{npz,json}: in_<i> (samples), out_<i> (the reference's stream), per case
codec / esz / ndims / ret, and whether the reference's decoder inverts its own stream."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from harness import Reference, gen_walk  # noqa: E402


def main():
    ref = Reference()
    rng = np.random.default_rng(20260929)
    out, cases = {}, []
    for ndims in (513, 600, 1000, 2047):
        for esz in (1, 2):
            n = ndims * 40 + 7                          # two groups, a block, a ragged tail
            data = gen_walk(rng, n, ndims, esz, 8, flat_every=3)
            for codec in ("delta", "xff"):
                buf, ret = ref.compress_raw(codec, data, ndims)
                nbytes = ret * esz
                # the stream's byte length: ret is in elements (floor); an odd byte count of a 16-bit stream shows in the next byte not being the fill
                if esz == 2 and buf[nbytes] != 0xAB:
                    nbytes += 1
                stream = buf[:nbytes].copy()
                dec, dret = ref.decompress(codec, stream, esz, n, ndims_hint=ndims)
                idx = len(cases)
                out[f"in_{idx}"] = data
                out[f"out_{idx}"] = stream
                cases.append(dict(idx=idx, codec=codec, esz=esz, ndims=ndims, n=int(n), ret=int(ret), nbytes=int(nbytes),
                                  ref_roundtrips=bool(dret == n and np.array_equal(dec[:n], data.ravel()))))
    gdir = os.path.join(ROOT, "tests", "golden")
    np.savez_compressed(os.path.join(gdir, "golden_wide_v1.npz"), **out)
    with open(os.path.join(gdir, "golden_wide_v1.json"), "w") as f:
        json.dump(dict(source="compiled reference dblalock/sprintz cpp/Compress via oracle/_ref: sprintz_compress_* at 513 .. 2047 columns",
                       generator="oracle/gen_golden_wide.py", cases=cases), f, indent=0)
    print(f"wrote {len(cases)} cases;", sum(c["ref_roundtrips"] for c in cases), "round-trip in the reference;",
          os.path.getsize(os.path.join(gdir, "golden_wide_v1.npz")), "bytes")


if __name__ == "__main__":
    main()

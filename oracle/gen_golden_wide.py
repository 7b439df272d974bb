#!/usr/bin/env python3
"""(v2: 2 048 .. 65 535 columns -- run with the argument v2 -- next to v1's 513 .. 2 047.)  Streams of 513 .. 65 535 columns minted by the compiled REFERENCE (make -C oracle ref): the reference's own tests stop at 129 columns
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
    v2 = len(sys.argv) > 1 and sys.argv[1] == "v2"
    name = "golden_wide_v2" if v2 else "golden_wide_v1"
    ref = Reference()
    rng = np.random.default_rng(20260930 if v2 else 20260929)
    out, cases = {}, []
    # v2.  What the format and the reference allow up there (probed with the compiled reference, 2026-09-29):
    #   * the header's remaining_len is a uint16 (format.h:36-45): a tail of more than 65 535 elements -- a spare block of >= 8 192 columns --
    #     is written TRUNCATED by the reference's encoder and cannot be decoded by anyone; the 8 192-column case pins exactly that
    #     (encoder bytes, and the length the reference's decoder returns for them);
    #   * the reference's decoder corrupts its heap from about 65 521 columns on (free(): corrupted unsorted chunks): the 65 535-column
    #     streams are minted by its encoder and its decoder is not run on them (ref_roundtrips: null).
    for ndims in ((2048, 4096, 8192, 65535) if v2 else (513, 600, 1000, 2047)):
        for esz in (1, 2):
            n = ndims * 40 + 7                          # two groups, a block, a ragged tail
            if ndims == 8192: n = ndims * 24 + 7        # one group and a block that does not fit remaining_len
            if ndims == 65535: n = ndims * 16 + 7       # one group, 7 elements of tail
            data = gen_walk(rng, n, ndims, esz, 8, flat_every=3)
            for codec in ("delta", "xff"):
                buf, ret = ref.compress_raw(codec, data, ndims)
                nbytes = ret * esz
                # the stream's byte length: ret is in elements (floor); an odd byte count of a 16-bit stream shows in the next byte not being the fill
                if esz == 2 and buf[nbytes] != 0xAB:
                    nbytes += 1
                stream = buf[:nbytes].copy()
                rt, dret = None, None
                if ndims < 65521:
                    dec, dret = ref.decompress(codec, stream, esz, n, ndims_hint=ndims)
                    rt = bool(dret == n and np.array_equal(dec[:n], data.ravel()))
                    assert dret <= n and np.array_equal(dec[:dret], data.ravel()[:dret]), "the reference's decoder returned something that is not a prefix"
                idx = len(cases)
                if codec == "delta" or not v2:
                    out[f"in_{idx}"] = data
                out[f"out_{idx}"] = stream
                cases.append(dict(idx=idx, codec=codec, esz=esz, ndims=ndims, n=int(n), ret=int(ret), nbytes=int(nbytes),
                                  in_idx=idx if (codec == "delta" or not v2) else idx - 1,      # (v2: the two codecs share their input)
                                  ref_roundtrips=rt, ref_dret=None if dret is None else int(dret)))
    gdir = os.path.join(ROOT, "tests", "golden")
    np.savez_compressed(os.path.join(gdir, name + ".npz"), **out)
    with open(os.path.join(gdir, name + ".json"), "w") as f:
        json.dump(dict(source="compiled reference dblalock/sprintz cpp/Compress via oracle/_ref: sprintz_compress_* at %s columns" % ("2048 .. 65535" if v2 else "513 .. 2047"),
                       generator="oracle/gen_golden_wide.py", cases=cases), f, indent=0)
    print(f"wrote {len(cases)} cases;", sum(bool(c["ref_roundtrips"]) for c in cases), "round-trip in the reference;",
          os.path.getsize(os.path.join(gdir, name + ".npz")), "bytes")


if __name__ == "__main__":
    main()

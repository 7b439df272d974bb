#!/usr/bin/env python3
"""What the REFERENCE DECODER actually outputs for the golden streams it does not invert (tests/golden/golden_v1: the cases with
ref_roundtrips == false -- sprintz_xff_rle.cpp:893-901 replays 16-bit FIRE runs with the coefficient shifted by 4 instead of
12), so that "the oracle's quirk mode == the reference decoder" is checked wherever the tests run, not only where
oracle/_ref exists.  Needs the compiled reference (make -C oracle ref); writes tests/golden/golden_refdec_v1.{npz,json}."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from harness import Reference  # noqa: E402


def main():
    gdir = os.path.join(ROOT, "tests", "golden")
    manifest = json.load(open(os.path.join(gdir, "golden_v1.json")))["cases"]
    arrays = np.load(os.path.join(gdir, "golden_v1.npz"))
    ref = Reference()
    out, cases = {}, []
    for m in manifest:
        if m["ref_roundtrips"]:
            continue
        data, stream = arrays[f"in_{m['idx']}"], arrays[f"out_{m['idx']}"]
        dec, dret = ref.decompress(m["codec"], stream, m["esz"], data.size, m["ndims"])
        assert dret == m["dec_ret"]
        out[f"refdec_{m['idx']}"] = dec
        cases.append(dict(idx=m["idx"], name=m["name"], dec_ret=int(dret), differs_from_input_at=[int(i) for i in np.flatnonzero(dec != data.ravel())[:8]]))
    np.savez_compressed(os.path.join(gdir, "golden_refdec_v1.npz"), **out)
    with open(os.path.join(gdir, "golden_refdec_v1.json"), "w") as f:
        json.dump(dict(source="compiled reference dblalock/sprintz cpp/Compress via oracle/_ref: sprintz_decompress_* on golden_v1's streams",
                       generator="oracle/gen_golden_refdec.py", cases=cases), f, indent=0)
    print(f"wrote {len(cases)} reference-decoder outputs:", [c["name"] for c in cases])


if __name__ == "__main__":
    main()

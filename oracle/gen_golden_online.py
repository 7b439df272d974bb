#!/usr/bin/env python3
"""Mint golden vectors for the reference's 2020 "online" u16 coders (online.hpp:395-445) from the
COMPILED REFERENCE.  TEST INFRASTRUCTURE ONLY; run in the build container after `make -C oracle ref`:

    python oracle/gen_golden_online.py

Writes tests/golden/golden_online_v1.npz (+ .json): input, the container the reference's *_pack_u16 wrote,
its return value (elements), and a mask of the container bytes the reference actually WRITES (it leaves header
padding and, for inputs of 0/1 elements, the choice bytes untouched: found by running it over two differently
filled buffers; such bytes are recorded as 0 and are 0 in our output).  Only data is stored."""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from harness import REF_SO, gen_fuzz, gen_known, gen_sparse, gen_walk  # noqa: E402

KINDS = ["dyndelta", "dyndelta_alt", "zigzag", "pack", "pack_zigzag"]


def main():
    ref = C.CDLL(REF_SO)
    ref.ref_online_pack.restype = C.c_int64
    ref.ref_online_pack.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_void_p]
    ref.ref_online_unpack.restype = C.c_int64
    ref.ref_online_unpack.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(20260928)
    arrays, manifest = {}, []
    sizes = [0, 1, 2, 3, 7, 8, 9, 10, 15, 16, 17, 18, 24, 25, 63, 64, 65, 66, 71, 72, 73, 127, 128, 129, 136, 137, 1000, 4096, 4113]   # compress_testing.hpp:452-463 + the edges of the 1 + 8 b framing
    idx = 0
    for kind in range(5):
        for sz in sizes:
            ins = [("zeros", np.zeros(sz, np.uint16)), ("known", gen_known(sz, 2)), ("max", np.full(sz, 65535, np.uint16)),
                   ("ramp", (np.arange(sz) * 7).astype(np.uint16)), ("squares", (np.arange(sz) ** 2).astype(np.uint16)),
                   ("alt", np.where(np.arange(sz) % 2 == 0, 0, 65535).astype(np.uint16)),
                   ("sparse", gen_sparse(rng, sz, 2, 0.1)), ("walk5", gen_walk(rng, sz, 1, 2, 5)), ("walk300", gen_walk(rng, sz, 1, 2, 300))]
            ins += [(f"fuzz{sh}", gen_fuzz(rng, sz, 2, sh)) for sh in (0, 2, 5, 9, 12, 15)]
            for tag, x in ins:
                x = np.ascontiguousarray(x)
                cap = sz * 2 + sz // 8 + 256
                a, b = np.zeros(cap, np.uint8), np.full(cap, 0xFF, np.uint8)
                ra = ref.ref_online_pack(kind, x.ctypes.data, sz, a.ctypes.data)
                rb = ref.ref_online_pack(kind, x.ctypes.data, sz, b.ctypes.data)
                assert ra == rb
                nbytes = int(ra) * 2
                defined = a[:nbytes] == b[:nbytes]
                back = np.zeros(sz + 16, np.uint16)
                dret = ref.ref_online_unpack(kind, a.ctypes.data, back.ctypes.data)
                assert dret == sz and np.array_equal(back[:sz], x), (kind, sz, tag)
                name = f"o{idx:04d}"
                arrays[name + "_in"] = x
                arrays[name + "_container"] = a[:nbytes].copy()
                arrays[name + "_defined"] = np.packbits(defined)
                manifest.append({"name": name, "kind": kind, "kind_name": KINDS[kind], "tag": tag, "n": sz, "ret": int(ra)})
                idx += 1
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "golden_online_v1.npz"), **arrays)
    with open(os.path.join(ROOT, "tests", "golden", "golden_online_v1.json"), "w") as f:
        json.dump({"version": 1, "source": "compiled reference: online.cpp via oracle/ref_shim.cpp", "cases": manifest}, f, indent=0)
    print(len(manifest), "cases")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Mint golden vectors from the COMPILED REFERENCE (oracle/_ref/libsprintz_ref.so).

TEST INFRASTRUCTURE ONLY.  Run in the build container (where /root/reference
exists, `make -C oracle ref` has been run):

    python oracle/gen_golden.py

Writes tests/golden/golden_v1.npz (+ golden_v1.json manifest): for each case the
input samples, the exact compressed byte stream the reference produced, and the
reference's return values.  The reference has no golden bytes of its own
(its tests are round-trip only), so these are the parity
contract.  Only data is stored -- no reference code.

The exact byte length of a reference stream is not returned by the reference
for 16-bit data (element-count return floors odd lengths,
sprintz_xff_rle.cpp:554); it is recovered here by compressing twice into
buffers with different poison bytes and taking the longest prefix that agrees
and whose element count matches the return value.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from harness import DTYPES, Reference, gen_fuzz, gen_known, gen_sparse, gen_walk  # noqa: E402


def ref_stream(ref, orc, codec, data, ndims):
    """(output buffer, ret, exact stream byte length) -- see module doc.

    The reference only returns an element count, which floors an odd byte
    length for 16-bit data, and its 16-bit low-dim encoder zeroes bytes past
    the payload (memset of data_sz*elem_sz, sprintz_delta_lowdim.cpp:303), so
    neither the return value nor "bytes both runs wrote" gives the length.
    The stream is self-framing, so the length is the number of bytes its
    framing spans (walked by oracle_decompress_ex); it must be 2*ret or 2*ret+1
    and every byte in it must have been written identically by two runs into
    differently poisoned buffers."""
    esz = data.dtype.itemsize
    buf_a, ret = ref.compress_raw(codec, data, ndims)
    src = np.concatenate([data.ravel(), np.zeros(64, data.dtype)])
    buf_b = np.full(buf_a.size, 0x5C, dtype=np.uint8)
    ret_b = ref._compress(0 if codec == "delta" else 1, esz, src.ctypes.data, data.size,
                          buf_b.ctypes.data, ndims, 1)
    assert ret == ret_b
    nbytes = orc.stream_nbytes(codec, buf_a, esz, data.size)
    assert nbytes // esz == ret, (nbytes, ret)
    assert np.array_equal(buf_a[:nbytes], buf_b[:nbytes])
    return buf_a, ret, nbytes


def cases():
    rng = np.random.default_rng(20240928)
    out = []

    def add(name, codec, esz, ndims, data):
        out.append(dict(name=name, codec=codec, esz=esz, ndims=ndims,
                        data=np.ascontiguousarray(data, dtype=DTYPES[esz]).ravel()))

    for esz in (1, 2):
        top = 1 << (8 * esz)
        for codec in ("delta", "xff"):
            for D in (1, 2, 3, 4, 5, 8, 16, 17, 32, 80):
                grp = 16 * D
                # tiny (< 128 elements -> raw) and ngroups == 0 via the loop guard
                add("tiny", codec, esz, D, gen_known(100, esz))
                if grp > 128:
                    add("ngroups0", codec, esz, D, gen_fuzz(rng, grp - 1, esz, 3))
                # whole groups + ragged tail, several magnitudes (exercises 7->8 / 15->16)
                for sh in ((0, 1, 2, 5) if esz == 1 else (0, 1, 2, 8, 9, 10, 13)):
                    add(f"fuzz_sh{sh}", codec, esz, D, gen_fuzz(rng, 6 * grp + 5, esz, sh))
                add("walk8", codec, esz, D, gen_walk(rng, 12 * grp + 3, D, esz, 8))
                add("walk_flat", codec, esz, D, gen_walk(rng, 40 * grp, D, esz, 30, flat_every=2))
                add("sparse", codec, esz, D, gen_sparse(rng, 10 * grp + 1, esz, 0.01))
                # |delta| = 40 -> zigzag 79/80 -> 7 bits -> header nibble 8 (Appendix B.2)
                rows = 48
                x = (np.arange(rows)[:, None] * 40 + np.arange(D)[None, :]) % top
                add("delta40", codec, esz, D, x)
                # all zeros, exactly 8 blocks: run ends at last_full_group_start (B.3)
                add("zeros_64rows", codec, esz, D, np.zeros(64 * D))
                add("zeros_65rows", codec, esz, D, np.zeros(65 * D))
                # run closing a group: nonzero block, zeros, nonzero
                x = np.zeros((8 * 9, D), np.int64)
                x[:8] = rng.integers(0, 50, (8, D))
                x[8:] = x[7]
                x[48:] = rng.integers(0, 50, (24, D))
                add("run_closes_group", codec, esz, D, x)
                x = np.zeros((8 * 10, D), np.int64)
                x[:16] = rng.integers(0, 50, (16, D))
                x[16:] = x[15]
                x[56:] = rng.integers(0, 50, (24, D))
                add("run_in_slot0", codec, esz, D, x)
            # long runs -> 2-byte varint; only for a few D to keep the file small
            for D in (1, 5, 8):
                x = np.zeros((8 * 140 + 24, D), np.int64)
                x[-24:] = rng.integers(0, 9, (24, D))
                add("run_gt127", codec, esz, D, x)
                x = np.full((8 * 300, D), 7, np.int64)
                add("const_300blocks", codec, esz, D, x)
    # high-variance low-dim 16-bit FIRE (Appendix B.4: 32-bit multiply)
    for D in (1, 2):
        x = (np.arange(4096)[:, None] % 2) * 65535
        add("lowdim16_extreme", "xff", 2, D, np.repeat(x, D, axis=1))
    # oscillation then FIRE-predicted decay: stream on which the REFERENCE DECODER
    # does not round-trip (DESIGN.md "Reference decoder quirk"); encoder bytes are golden.
    D = 3
    v = 1000
    rows = []
    for i in range(8):
        v += 100 if i % 2 == 0 else -100
        rows.append([v] * D)
    for dl in [6, -1, 0, 0, 0, 0, 0, 0]:
        v += dl
        rows.append([v] * D)
    rows += [[v] * D] * 48
    rows += [list(rng.integers(0, 50, D) + v) for _ in range(32)]
    add("fire16_run_nonzero_pred", "xff", 2, D, np.array(rows))
    return out


def main():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from harness import Oracle
    ref = Reference()
    orc = Oracle()
    arrays = {}
    manifest = []
    for i, c in enumerate(cases()):
        data = c["data"]
        buf, ret, nbytes = ref_stream(ref, orc, c["codec"], data, c["ndims"])
        stream = buf[:nbytes].copy()
        # generation-time sanity check only: our restatement already agrees
        so, ro = orc.compress(c["codec"], data, c["ndims"])
        assert ro == ret and np.array_equal(stream, so), (c["name"], ro, ret, len(so), nbytes)
        dec, dret = ref.decompress(c["codec"], stream, c["esz"], data.size, c["ndims"])
        ref_roundtrips = bool(dret == data.size and np.array_equal(dec, data))
        arrays[f"in_{i}"] = data
        arrays[f"out_{i}"] = stream
        manifest.append(dict(idx=i, name=c["name"], codec=c["codec"], esz=c["esz"], ndims=c["ndims"],
                             n=int(data.size), nbytes=int(stream.size), ret=int(ret),
                             dec_ret=int(dret), ref_roundtrips=ref_roundtrips))
    gdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(gdir, exist_ok=True)
    np.savez_compressed(os.path.join(gdir, "golden_v1.npz"), **arrays)
    with open(os.path.join(gdir, "golden_v1.json"), "w") as f:
        json.dump(dict(source="compiled reference dblalock/sprintz cpp/Compress via oracle/_ref",
                       generator="oracle/gen_golden.py", cases=manifest), f, indent=0)
    nbad = sum(not m["ref_roundtrips"] for m in manifest)
    print(f"wrote {len(manifest)} cases; reference decoder fails to round-trip {nbad} of them")
    print("npz bytes:", os.path.getsize(os.path.join(gdir, "golden_v1.npz")))


if __name__ == "__main__":
    main()

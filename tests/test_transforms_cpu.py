"""Stand-alone transforms, CPU side: the oracle (oracle/transforms_oracle.c) against containers
minted from the compiled reference, and against the reference itself where it was built."""
import numpy as np

from harness import DTYPES


def test_oracle_matches_reference_containers(oracle, golden_transforms):
    manifest, arrays = golden_transforms
    assert len(manifest) >= 300
    for m in manifest:
        x, cont = arrays[m["name"] + "_in"], arrays[m["name"] + "_container"]
        got, ret = oracle.transform_encode(m["kind"], x, m["ndims"])
        assert ret == m["ret"] and np.array_equal(got, cont), m
        back, dret = oracle.transform_decode(m["kind"], cont, m["esz"])
        assert dret == m["n"] and np.array_equal(back, x), m


def test_oracle_vs_compiled_reference(oracle, reference):
    import pytest
    if not reference.has_transforms():
        pytest.skip("oracle/_ref built before the transform shim was added")
    rng = np.random.default_rng(3)
    for esz in (1, 2):
        for kind in (0, 1, 2):
            for D in (1, 4, 7, 16, 31, 32, 65, 300):
                for n in (1, 5, D, 2 * D - 1, 8 * D + 1, 777, 5000, 24 * D + 3):
                    x = rng.integers(0, 1 << (8 * esz), n).astype(DTYPES[esz])
                    a, ra = oracle.transform_encode(kind, x, D)
                    b, rb = reference.transform_encode(kind, x, D)
                    assert ra == rb and np.array_equal(a, b), (esz, kind, D, n)
                    da, _ = oracle.transform_decode(kind, a, esz)
                    db, _ = reference.transform_decode(kind, a, esz)
                    assert np.array_equal(da, x) and np.array_equal(db, x), (esz, kind, D, n)


def test_definition(oracle):
    """per column: y[r] = x[r] - x[r-1]  /  x[r] - 2 x[r-1] + x[r-2], zero initial state, wrapping"""
    rng = np.random.default_rng(4)
    for esz in (1, 2):
        x = rng.integers(0, 1 << (8 * esz), (50, 6)).astype(np.int64)
        z = np.zeros((2, 6), np.int64)
        xp = np.vstack([z, x])
        for kind, want in ((0, xp[2:] - xp[1:-1]), (1, xp[2:] - 2 * xp[1:-1] + xp[:-2])):
            cont, _ = oracle.transform_encode(kind, x.astype(DTYPES[esz]).reshape(-1), 6)
            got = cont[6:].view(DTYPES[esz])
            assert np.array_equal(got, (want % (1 << (8 * esz))).astype(DTYPES[esz]).reshape(-1))

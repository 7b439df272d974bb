"""CPU tests: the oracle (our C restatement) against golden vectors minted from
the compiled reference (tests/golden, generator oracle/gen_golden.py).
Bit-exact: stream bytes, return values, decoded samples."""
import numpy as np


def test_golden_manifest_covers_matrix(golden):
    manifest, _ = golden
    combos = {(m["codec"], m["esz"], m["ndims"]) for m in manifest}
    for codec in ("delta", "xff"):
        for esz in (1, 2):
            for D in (1, 2, 3, 4, 5, 8, 16, 17, 32, 80):
                assert (codec, esz, D) in combos
    names = {m["name"] for m in manifest}
    for need in ("tiny", "ngroups0", "zeros_64rows", "run_closes_group", "run_gt127",
                 "delta40", "fire16_run_nonzero_pred"):
        assert need in names
    # odd compressed byte lengths for 16-bit streams are present (floor'ed return)
    assert any(m["esz"] == 2 and m["nbytes"] % 2 == 1 for m in manifest)


def test_oracle_encoder_matches_golden(oracle, golden):
    manifest, arrays = golden
    for m in manifest:
        data = arrays[f"in_{m['idx']}"]
        want = arrays[f"out_{m['idx']}"]
        got, ret = oracle.compress(m["codec"], data, m["ndims"])
        assert ret == m["ret"], m
        assert got.size == m["nbytes"] and np.array_equal(got, want), m


def test_oracle_decoder_inverts_golden(oracle, golden):
    manifest, arrays = golden
    for m in manifest:
        data = arrays[f"in_{m['idx']}"]
        stream = arrays[f"out_{m['idx']}"]
        dec, ret = oracle.decompress(m["codec"], stream, m["esz"], data.size)
        assert ret == data.size, m
        assert np.array_equal(dec, data), m
        if m["ref_roundtrips"]:
            assert ret == m["dec_ret"]


def test_reference_decoder_quirk_is_modelled(oracle, golden):
    """The one golden stream the reference decoder does not invert
    (sprintz_xff_rle.cpp:894-901): our default decoder is lossless."""
    manifest, arrays = golden
    bad = [m for m in manifest if not m["ref_roundtrips"]]
    assert [m["name"] for m in bad] == ["fire16_run_nonzero_pred"]
    m = bad[0]
    data = arrays[f"in_{m['idx']}"]
    stream = arrays[f"out_{m['idx']}"]
    dec, _ = oracle.decompress("xff", stream, 2, data.size)
    assert np.array_equal(dec, data)
    quirk, qret = oracle.decompress("xff", stream, 2, data.size, quirk=1)
    assert not np.array_equal(quirk, data)
    # ... and the quirk mode IS the reference decoder: its actual output for this stream, stored when the fixtures were minted
    # (oracle/gen_golden_refdec.py), so this holds wherever the tests run -- not only where oracle/_ref exists
    import json
    import os
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    cases = json.load(open(os.path.join(gdir, "golden_refdec_v1.json")))["cases"]
    refdec = np.load(os.path.join(gdir, "golden_refdec_v1.npz"))
    assert [c["idx"] for c in cases] == [m["idx"]]
    assert qret == cases[0]["dec_ret"] == m["dec_ret"]
    assert np.array_equal(quirk, refdec[f"refdec_{m['idx']}"])


def test_oracle_at_513_to_2047_columns_is_the_reference(oracle, golden_wide):
    """golden_wide_v1: the compiled reference at widths its own tests never reach (compress_testing.hpp:20-21 stops at 129 columns);
    the oracle writes the same bytes and inverts them"""
    manifest, arrays = golden_wide
    assert {m["ndims"] for m in manifest} == {513, 600, 1000, 2047} and len(manifest) == 16
    for m in manifest:
        data, want = arrays[f"in_{m['idx']}"], arrays[f"out_{m['idx']}"]
        got, ret = oracle.compress(m["codec"], data, m["ndims"])
        assert ret == m["ret"] and got.size == m["nbytes"] and np.array_equal(got, want), m
        dec, dret = oracle.decompress(m["codec"], want, m["esz"], data.size)
        assert dret == data.size and np.array_equal(dec, data.ravel()), m


def test_oracle_at_2048_to_65535_columns_is_the_reference(oracle, golden_wide2):
    """golden_wide_v2: the header's ndims is a full uint16 (format.h:36-45).  The reference's ENCODER takes every width; the oracle writes its
    bytes.  Where the reference's decoder could be run (below ~65 521 columns it does not corrupt its heap) the oracle's decoder returns what
    it returned -- including the 8 192-column streams whose tail does not fit the header's uint16 remaining_len: truncated by the encoder,
    a decoded prefix for everybody."""
    manifest, arrays = golden_wide2
    assert {m["ndims"] for m in manifest} == {2048, 4096, 8192, 65535} and len(manifest) == 16
    for m in manifest:
        data, want = arrays[f"in_{m['in_idx']}"], arrays[f"out_{m['idx']}"]
        got, ret = oracle.compress(m["codec"], data, m["ndims"])
        assert ret == m["ret"] and got.size == m["nbytes"] and np.array_equal(got, want), m
        dec, dret = oracle.decompress(m["codec"], want, m["esz"], data.size)
        want_ret = m["ref_dret"] if m["ref_dret"] is not None else data.size
        assert dret == want_ret and np.array_equal(dec[:dret], data.ravel()[:dret]), m
        assert (m["ref_roundtrips"] is False) == (m["ndims"] == 8192), m


def test_compress_chunks_on_many_threads_is_compress_chunks(oracle):
    """harness.Oracle.compress_chunks_mt (the full-size GPU tests' all-chunk comparison) == the serial form, ragged last chunk too"""
    from harness import gen_walk
    rng = np.random.default_rng(77)
    data = gen_walk(rng, 37 * 5120 + 1234, 8, 2, 8, flat_every=3)
    serial = oracle.compress_chunks("xff", data, 5120, 8)
    for threads in (1, 3, 8):
        dest, stride, sizes = oracle.compress_chunks_mt("xff", data, 5120, 8, threads=threads)
        assert sizes.tolist() == [s.size for s in serial]
        assert all(np.array_equal(dest[c * stride:c * stride + s.size], s) for c, s in enumerate(serial))

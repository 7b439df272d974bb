"""Dataset plumbing (sprintz_amd/datasets.py; reference: python/datasets/compress_bench.py:45-120)."""
import numpy as np
import pytest

from sprintz_amd import datasets


def test_quantize_matches_the_reference_recipe():
    rng = np.random.default_rng(0)
    mat = rng.normal(size=(500, 4)) * [1, 10, 100, 0] + [5, -3, 0, 7]      # the last variable is constant
    for dt, top in ((np.uint8, 255), (np.uint16, 65535)):
        q = datasets.quantize(mat, dt)
        assert q.dtype == dt and q.shape == mat.shape
        m = mat - mat.min(axis=0, keepdims=True)
        m = m.astype(np.float32)
        m /= np.maximum(1, m.max(axis=0, keepdims=True))
        assert np.array_equal(q, (m * top).astype(dt))
        assert q[:, :3].max(axis=0).tolist() == [top, top, top] and (q[:, 3] == 0).all()
    with pytest.raises(ValueError):
        datasets.quantize(mat, np.int32)


def test_dump_and_load_both_orders(tmp_path):
    rng = np.random.default_rng(1)
    q = rng.integers(0, 65536, (1000, 6)).astype(np.uint16)
    pc = datasets.dump(q, str(tmp_path / "rowmajor.dat"), "c")
    pf = datasets.dump(q, str(tmp_path / "colmajor.dat"), "f")
    assert np.array_equal(datasets.load(pc, np.uint16, 6, "c"), q)
    assert np.array_equal(datasets.load(pf, np.uint16, 6, "f"), q.T)
    assert np.array_equal(np.fromfile(pf, np.uint16)[:1000], q[:, 0])      # a variable's samples are contiguous
    with pytest.raises(ValueError):
        datasets.load(pc, np.uint16, 7)


@pytest.mark.gpu
def test_files_compress_identically_in_either_layout(tmp_path):
    import torch
    rng = np.random.default_rng(2)
    mat = np.cumsum(rng.normal(size=(20000, 8)), axis=0)
    q = datasets.quantize(mat, np.uint16)
    pc = datasets.dump(q, str(tmp_path / "r.dat"), "c")
    pf = datasets.dump(q, str(tmp_path / "c.dat"), "f")
    cd, b_row = datasets.compress_file(pc, np.uint16, 8, "c")
    _, b_col = datasets.compress_file(pf, np.uint16, 8, "f")
    assert torch.equal(b_row.sizes, b_col.sizes)
    assert torch.equal(b_row.data[: b_row.total_bytes()], b_col.data[: b_col.total_bytes()])
    assert np.array_equal(cd.decompress(b_row).cpu().numpy().view(np.uint16).reshape(-1, 8), q)
    assert np.array_equal(cd.decompress_colmajor(b_col).cpu().numpy().view(np.uint16), q.T)


# ---------------------------------------------------------------- real (measured) data: what is in the image without a network
# (sprintz_amd.datasets.offline_real_datasets: scikit-learn's bundled tabular sets and photographs, quantised as compress_bench.py:45-60 does)

def _real_cases():
    cases = []
    for name, mat in datasets.offline_real_datasets():
        for dt in (np.uint8, np.uint16):
            cases.append((f"{name}-u{8 * np.dtype(dt).itemsize}", datasets.quantize(mat, dt)))
    return cases


def test_real_data_is_present_and_round_trips_through_the_oracle(oracle):
    cases = _real_cases()
    if not cases:
        pytest.skip("scikit-learn's bundled datasets are not importable here")
    assert len(cases) >= 12
    for name, q in cases:
        ndims = q.shape[1]
        for codec in ("delta", "xff"):
            comp, ret = oracle.compress(codec, q, ndims)
            assert ret == comp.size // q.dtype.itemsize, (name, codec)
            dec, dret = oracle.decompress(codec, comp, q.dtype.itemsize, q.size)
            assert dret == q.size and np.array_equal(dec, q.ravel()), (name, codec)


def test_real_data_reference_and_oracle_write_the_same_bytes(oracle, reference):
    """the pin, on measured data: the compiled reference (where oracle/_ref exists) and the restatement, stream for stream -- up to the
    photographs' 1 920 columns (the reference's own tests stop at 129, compress_testing.hpp:20-21)"""
    cases = _real_cases()
    if not cases:
        pytest.skip("scikit-learn's bundled datasets are not importable here")
    for name, q in cases:
        ndims = q.shape[1]
        for codec in ("delta", "xff"):
            buf, wret = reference.compress_raw(codec, q, ndims)
            got, ret = oracle.compress(codec, q, ndims)
            assert ret == wret and np.array_equal(buf[:got.size], got), (name, codec)


@pytest.mark.gpu
def test_real_data_on_the_gpu_is_the_oracle_byte_for_byte(oracle):
    """every real data set, 8 and 16 bits, both codecs: the single-call encoder writes the oracle's stream, the decoder returns the
    samples; in 10 KB chunks (row-major and column-major sources) every chunk's stream is the oracle's and the batch decodes back"""
    import torch
    import sprintz_amd as sz
    cases = _real_cases()
    if not cases:
        pytest.skip("scikit-learn's bundled datasets are not importable here")
    for name, q in cases:
        ndims, esz = q.shape[1], q.dtype.itemsize
        for codec in ("delta", "xff"):
            want, wret = oracle.compress(codec, q, ndims)
            if q.size <= (1 << 20):                       # one call, host pointers (the drop-in symbols)
                fn = getattr(sz, f"sprintz_compress_{codec}_{8 * esz}b")
                dest = np.zeros(q.size * 3 // 2 + 64, np.int8 if esz == 1 else np.int16)
                ret = fn(np.ascontiguousarray(q), q.size, dest, ndims)
                assert ret == wret and np.array_equal(dest.view(np.uint8)[:want.size], want), (name, codec, "single call")
                back = np.zeros(q.size + 64, q.dtype)
                dret = getattr(sz, f"sprintz_decompress_{codec}_{8 * esz}b")(dest, back)
                assert dret == q.size and np.array_equal(back[:q.size], q.ravel()), (name, codec, "single call decode")
            rows = max(64, (10240 // (ndims * esz)) // 8 * 8)      # ~10 KB of rows, at least a few stream groups of a wide set
            rows = min(rows, max(8, q.shape[0] // 8 * 8))
            nrows = q.shape[0] // rows * rows             # whole chunks (the column-major entry points take whole chunks)
            if nrows == 0:
                continue
            qq = np.ascontiguousarray(q[:nrows])
            cd = sz.ChunkedCodec(codec, esz, ndims, rows * ndims, device="cuda:0")
            t = torch.from_numpy(qq.view(np.int8 if esz == 1 else np.int16)).cuda().view(cd.dtype)
            b = cd.compress(t)
            streams = oracle.compress_chunks(codec, qq, rows * ndims, ndims)
            data, offs, sizes = b.data.cpu().numpy(), b.offsets.cpu().numpy(), b.sizes.cpu().numpy()
            for c, s in enumerate(streams):
                assert sizes[c] == s.size and np.array_equal(data[offs[c]:offs[c] + sizes[c]], s), (name, codec, "chunk", c)
            assert np.array_equal(cd.decompress(b).cpu().numpy().view(q.dtype).reshape(-1, ndims), qq), (name, codec, "batch decode")
            if ndims <= 512 and rows % 8 == 0:
                tc = torch.from_numpy(np.ascontiguousarray(qq.T).view(np.int8 if esz == 1 else np.int16)).cuda().view(cd.dtype)
                bc = cd.compress_colmajor(tc)
                assert torch.equal(bc.sizes, b.sizes) and torch.equal(bc.data[: bc.total_bytes()], b.data[: b.total_bytes()]), (name, codec, "column-major source")
                assert np.array_equal(cd.decompress_colmajor(bc).cpu().numpy().view(q.dtype), qq.T), (name, codec, "column-major decode")

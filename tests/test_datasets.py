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

"""GPU tests (-m gpu): column-major matrices (BASELINE config 5).  The streams must be what
the oracle produces for the row-major flattening of each row range; decoding must restore
the column-major matrix."""
import zlib

import numpy as np
import pytest

from harness import DTYPES

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sz():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import sprintz_amd
    return sprintz_amd


CM_CONFIGS = [
    # name, codec, esz, ndims, rows_per_chunk, nrows, col_stride (0 = nrows)
    ("cfg5 u16 D=32 xff 160-row chunks (fast kernels)", "xff", 2, 32, 160, 160 * 41 + 57, 160 * 42),
    ("u16 D=8 xff 640-row chunks", "xff", 2, 8, 640, 640 * 20, 0),
    ("u16 D=8 delta", "delta", 2, 8, 640, 640 * 7 + 8, 640 * 8),
    ("u8 D=16 xff", "xff", 1, 16, 256, 256 * 9 + 100, 256 * 10 + 8),
    ("u8 D=64 delta", "delta", 1, 64, 128, 128 * 12, 0),
    ("u16 D=5 (generic kernels: odd column count)", "xff", 2, 5, 200, 200 * 6 + 33, 0),
    ("u16 D=8, unaligned stride (generic kernels)", "xff", 2, 8, 100, 100 * 9 + 3, 903),
    ("u8 D=2 low-dim", "delta", 1, 2, 512, 512 * 5 + 1, 0),
    ("u16 D=100 (2 columns per lane)", "xff", 2, 100, 64, 64 * 7 + 5, 64 * 8),
    ("u16 D=6 (fast kernels, group not full)", "xff", 2, 6, 160, 160 * 9 + 48, 160 * 10),
    ("u8 D=12 (fast kernels, group not full)", "delta", 1, 12, 256, 256 * 5 + 64, 256 * 6),
    ("u16 D=48 delta (fast kernels, group not full)", "delta", 2, 48, 104, 104 * 11 + 16, 104 * 12),
]


@pytest.mark.parametrize("enc_pair", [1, 0])          # two columns per lane (encode_wide.h, column-major source) / one (encode_fast.h)
@pytest.mark.parametrize("name,codec,esz,ndims,rpc,nrows,cs", CM_CONFIGS)
def test_colmajor_matches_oracle_on_the_transposed_view(sz, oracle, request, name, codec, esz, ndims, rpc, nrows, cs, enc_pair):
    import torch
    from sprintz_amd import _lib
    _lib.check(_lib.set_option(_lib.OPT_ENC_PAIR, enc_pair))
    request.addfinalizer(lambda: _lib.set_option(_lib.OPT_ENC_PAIR, 1))
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    cs = cs or nrows
    top = 1 << (8 * esz)
    rows = (np.cumsum(rng.integers(-6, 7, (nrows, ndims)), axis=0) % top).astype(DTYPES[esz])    # [nrows, ndims] row-major
    rows[nrows // 3: nrows // 3 + 2 * rpc + 5] = rows[nrows // 3]                                 # runs across chunk boundaries
    rows[-rpc // 2:] = rng.integers(0, top, (rpc // 2, ndims))                                    # incompressible end
    cols = np.full((ndims, cs), 0xEE, DTYPES[esz])
    cols[:, :nrows] = rows.T
    cd = sz.ChunkedCodec(codec, esz, ndims, rpc * ndims, device="cuda:0")
    ct = torch.from_numpy(cols.view(np.int8 if esz == 1 else np.int16)).cuda().view(cd.dtype)
    batch = cd.compress_colmajor(ct, nrows)
    want = oracle.compress_chunks(codec, rows.reshape(-1), rpc * ndims, ndims)
    comp, offs, sizes = batch.data.cpu().numpy(), batch.offsets.cpu().numpy(), batch.sizes.cpu().numpy()
    assert batch.nchunks == len(want)
    for c in range(batch.nchunks):
        assert sizes[c] == want[c].size, (name, c)
        assert np.array_equal(comp[offs[c]:offs[c] + sizes[c]], want[c]), (name, c)
    # the container is interchangeable with the row-major entry points
    out_rm = cd.decompress(batch)
    assert np.array_equal(out_rm.cpu().numpy().view(DTYPES[esz]), rows.reshape(-1)), name
    # and decodes back into a column-major matrix, touching nothing else
    nch = batch.nchunks
    out = torch.full((ndims, nch * rpc + 8), 0x77, dtype=torch.uint8, device="cuda:0").to(cd.dtype) if esz == 1 else \
        torch.full((ndims, nch * rpc + 8), 0x7777, dtype=torch.int32, device="cuda:0").to(torch.uint16)
    got = cd.decompress_colmajor(batch, out=out)
    g = got.cpu().numpy().view(DTYPES[esz])
    assert np.array_equal(g, rows.T), name
    rest = out[:, nrows:].cpu().numpy().view(DTYPES[esz])
    assert (rest == (0x77 if esz == 1 else 0x7777)).all(), name


def test_cfg5_full_shape(sz):
    """1 048 576 rows x 32 variables, uint16, column-major, 160-row chunks: round trip and
    agreement with the row-major path on the transposed data"""
    import torch
    nrows, D, rpc = 1 << 20, 32, 160
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    steps = torch.randint(-8, 9, (D, nrows), generator=g, device="cuda", dtype=torch.int32)
    cols = (torch.cumsum(steps, dim=1) & 0xffff).to(torch.uint16).contiguous()
    cd = sz.ChunkedCodec("xff", 2, D, rpc * D)
    batch = cd.compress_colmajor(cols)
    rm = cols.view(torch.int16).t().contiguous().view(torch.uint16)
    batch_rm = cd.compress(rm)
    assert torch.equal(batch.sizes, batch_rm.sizes)
    assert torch.equal(batch.data[: batch.total_bytes()], batch_rm.data[: batch_rm.total_bytes()])
    back = cd.decompress_colmajor(batch)
    assert torch.equal(back, cols)

"""GPU tests (-m gpu): the reference's non-RLE codecs (compress_rowmajor[_delta]_{8b,16b}) through
the C-ABI, against streams minted from the compiled reference and the oracle.  Nothing here
reads /root/reference."""
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


def test_reference_streams_single_call(sz, golden_norle):
    manifest, arrays = golden_norle
    for m in manifest[::2]:
        x, stream = arrays[m["name"] + "_in"], arrays[m["name"] + "_stream"]
        esz, D, n = m["esz"], m["ndims"], m["n"]
        if m["raw"] == 2:
            comp, dec = sz.compress8b_rowmajor_xff, sz.decompress8b_rowmajor_xff
        else:
            name = ("" if m["raw"] else "delta_") + f"{8 * esz}b"
            comp = getattr(sz, f"compress_rowmajor_{name}")
            dec = getattr(sz, f"decompress_rowmajor_{name}")
        dest = np.full(stream.size + 64 + 4 * D, 0xAB, np.uint8)
        ret = comp(x, n, dest, D)
        assert ret == m["ret"], m
        assert np.array_equal(dest[:stream.size], stream), m
        out = np.full(n + 32, 0xCD, DTYPES[esz])
        assert dec(np.concatenate([stream, np.zeros(32, np.uint8)]), out) == n, m
        assert np.array_equal(out[:n], x) and (out[n:] == 0xCD).all(), m


@pytest.mark.parametrize("codec,esz,ndims,chunk_len", [("delta_norle", 2, 8, 5120), ("bitpack", 2, 8, 5120), ("delta_norle", 1, 80, 10240),
                                                       ("bitpack", 1, 3, 999), ("delta_norle", 2, 300, 9600 + 31), ("delta_norle", 1, 1, 1024),
                                                       ("xff_norle", 1, 8, 4096), ("xff_norle", 1, 80, 10240)])
def test_batched_matches_oracle(sz, oracle, codec, esz, ndims, chunk_len):
    import torch
    rng = np.random.default_rng(zlib.crc32(f"{codec}{esz}{ndims}".encode()))
    nchunks = 70
    n = nchunks * chunk_len - chunk_len // 3
    top = 1 << (8 * esz)
    x = (np.cumsum(rng.integers(-5, 6, n)) % top).astype(DTYPES[esz])
    x[n // 4: n // 4 + 2 * chunk_len] = 0                          # all-zero blocks: no payload, no run length
    x[n - chunk_len // 2:] = rng.integers(0, top, chunk_len // 2)
    raw = {"delta_norle": 0, "bitpack": 1, "xff_norle": 2}[codec]
    cd = sz.ChunkedCodec(codec, esz, ndims, chunk_len, device="cuda:0")
    batch = cd.compress(torch.from_numpy(x.view(np.int8 if esz == 1 else np.int16)).cuda().view(cd.dtype))
    comp, offs, sizes = batch.data.cpu().numpy(), batch.offsets.cpu().numpy(), batch.sizes.cpu().numpy()
    for c in range(nchunks):
        want, _ = oracle.compress_norle(raw, x[c * chunk_len:(c + 1) * chunk_len], ndims)
        assert sizes[c] == want.size, (codec, c)
        assert np.array_equal(comp[offs[c]:offs[c] + sizes[c]], want), (codec, c)
    rets = torch.empty(nchunks, dtype=torch.int64, device="cuda:0")
    out = cd.decompress(batch, rets=rets)
    assert np.array_equal(out.cpu().numpy().view(DTYPES[esz]), x), codec
    r = rets.cpu().numpy()
    assert (r[:-1] == chunk_len).all() and r[-1] == n - (nchunks - 1) * chunk_len
